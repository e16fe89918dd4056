#!/bin/bash
mkdir -p gpurun_out/r03r
timeout -k 10 200 python -m pytest tests/test_e2e_gpu.py -m gpu -q -k "chain_hiera" > gpurun_out/r03r/chain.log 2>&1; tail -2 gpurun_out/r03r/chain.log
for v in "" "MSAM2_GEMM_WSTAT=3" "MSAM2_G96_V1=1"; do
  echo "== train_full_bench $v"
  env $v timeout -k 10 300 python tools/train_full_bench.py 2>&1 | grep -v amdgpu | tail -3
done
