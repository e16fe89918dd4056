#!/bin/bash
timeout -k 10 400 python -m pytest tests/test_backward_gpu.py tests/test_kernels_gpu.py -m gpu -q -k "gemm_tt or four_wave or attention_vs_oracle" 2>&1 | tail -4
