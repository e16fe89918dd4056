#!/bin/bash
out=gpurun_out/r03g
mkdir -p $out
python -m pytest tests/test_autograd_gpu.py tests/test_config4_at_size_gpu.py -m gpu -q > $out/t1.log 2>&1; echo "autograd+config4 rc=$?"; tail -3 $out/t1.log | cut -c1-300
python -m pytest tests/test_bf16_build_gpu.py -m gpu -q -s > $out/bf16.log 2>&1; echo "bf16 rc=$?"; tail -25 $out/bf16.log | cut -c1-300
python -m pytest tests/test_e2e_gpu.py -m gpu -q -k "512_slices" > $out/t512.log 2>&1; echo "512 rc=$?"; tail -3 $out/t512.log | cut -c1-300
