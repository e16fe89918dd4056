#!/bin/bash
mkdir -p gpurun_out/r03s
timeout -k 10 600 python -m pytest tests/test_backward_gpu.py tests/test_backward_encoder_gpu.py -m gpu -q -x > gpurun_out/r03s/bwd.log 2>&1; echo "bwd rc=$?"; tail -3 gpurun_out/r03s/bwd.log
timeout -k 10 300 python tools/train_full_bench.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-120
timeout -k 10 300 python tools/memattn_bwd_bench.py 2>&1 | grep -v amdgpu | tail -2
