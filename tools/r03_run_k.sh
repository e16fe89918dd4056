#!/bin/bash
# round 3 batch K: the D = 96 64-queries-per-wave attention kernel and the polynomial GELU
mkdir -p gpurun_out/r03k
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention_vs_oracle or gelu or gemm_epilogue or ln_mlp or conv3x3 or layernorm" > gpurun_out/r03k/tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r03k/tests.log
timeout -k 10 200 python tools/g96_bench.py > gpurun_out/r03k/g96.txt 2>&1 && cat gpurun_out/r03k/g96.txt
timeout -k 10 200 python tools/wstat_bench.py > gpurun_out/r03k/wstat.txt 2>&1 && cat gpurun_out/r03k/wstat.txt
