#!/bin/bash
mkdir -p gpurun_out/r03u
timeout -k 10 300 python -m pytest tests/test_backward_gpu.py -m gpu -q -x -k "gemm_tt or linear_backward or mlp_backward" > gpurun_out/r03u/tt.log 2>&1; echo "tt rc=$?"; tail -3 gpurun_out/r03u/tt.log
echo "== new"; timeout -k 10 200 python tools/gemm_tt_bench.py 2>&1 | grep -v amdgpu
echo "== old"; MSAM2_GEMM_TT_V1=1 timeout -k 10 200 python tools/gemm_tt_bench.py 2>&1 | grep -v amdgpu
timeout -k 10 300 python tools/train_full_bench.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-100
