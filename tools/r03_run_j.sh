#!/bin/bash
out=gpurun_out/r03j
mkdir -p $out
python -m pytest tests/ -m gpu -q --deselect tests/test_bf16_build_gpu.py -k "not 512_slices" > $out/tests_all.log 2>&1; echo "tests rc=$?"; tail -6 $out/tests_all.log | cut -c1-300
python -m pytest tests/test_bf16_build_gpu.py -m gpu -q -s > $out/bf16.log 2>&1; echo "bf16 rc=$?"; tail -12 $out/bf16.log | cut -c1-300
python tools/memattn_bwd_bench.py > $out/memattn_bwd.txt 2>&1; tail -5 $out/memattn_bwd.txt | cut -c1-250
