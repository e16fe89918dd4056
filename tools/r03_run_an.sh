#!/bin/bash
MSAM2_LIB_PATH=$PWD/medical-sam2_amd/libmsam2_hip_bf16.so timeout -k 10 600 python -m pytest tests/test_backward_encoder_gpu.py tests/test_grads_golden.py tests/test_backward_gpu.py -m gpu -q 2>&1 | tail -3
