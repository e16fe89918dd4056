#!/bin/bash
python -m pytest tests/test_bf16_build_gpu.py -m gpu -q 2>&1 | tail -3; tail -2 gpurun_out/bf16_suite.log
