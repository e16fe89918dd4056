#!/bin/bash
# round-3 GPU batch B: autograd bridge, config-3 / 512-slice / configs[4] tests, then the full bf16 suite (to see what it needs)
out=gpurun_out/r03b
mkdir -p $out
python -m pytest tests/test_autograd_gpu.py -m gpu -x -q -s > $out/autograd.log 2>&1; echo "autograd rc=$?"; tail -15 $out/autograd.log
python -m pytest tests/test_e2e_gpu.py -m gpu -q -s -k "config3 or configs3" > $out/config3.log 2>&1; echo "config3 rc=$?"; tail -8 $out/config3.log
python -m pytest tests/test_config4_at_size_gpu.py -m gpu -q -s > $out/config4.log 2>&1; echo "config4 rc=$?"; tail -12 $out/config4.log
python -m pytest tests/test_bf16_build_gpu.py -m gpu -q -s > $out/bf16.log 2>&1; echo "bf16 rc=$?"; tail -30 $out/bf16.log
