#!/bin/bash
mkdir -p gpurun_out/r03aa
timeout -k 10 600 python -m pytest tests/test_backward_gpu.py tests/test_backward_encoder_gpu.py tests/test_grads_golden.py tests/test_bptt_gpu.py tests/test_autograd_gpu.py -m gpu -q > gpurun_out/r03aa/bwd.log 2>&1; echo "bwd rc=$?"; tail -4 gpurun_out/r03aa/bwd.log
timeout -k 10 300 python tools/train_full_bench.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-100
