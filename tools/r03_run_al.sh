#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_backward_encoder_gpu.py tests/test_grads_golden.py tests/test_config4_at_size_gpu.py -m gpu -q 2>&1 | tail -3
timeout -k 10 300 python tools/train_full_bench.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-60
