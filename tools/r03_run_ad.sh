#!/bin/bash
mkdir -p gpurun_out/r03ad
timeout -k 10 500 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-bf16 --no-rooflines --slices 32 > gpurun_out/r03ad/n2.json 2> gpurun_out/r03ad/n2.err; echo "n2 rc=$?"; cut -c1-400 gpurun_out/r03ad/n2.json; tail -3 gpurun_out/r03ad/n2.err
timeout -k 10 500 python bench.py --gpus 2 --mode volume --slices 32 --steps 1 --warmup 1 > gpurun_out/r03ad/vol2.json 2> gpurun_out/r03ad/vol2.err; echo "vol2 rc=$?"; cut -c1-500 gpurun_out/r03ad/vol2.json; tail -3 gpurun_out/r03ad/vol2.err
