#!/bin/bash
for cfg in "MSAM2_TT_WGS=128" "MSAM2_TT_WGS=192" "MSAM2_TT_WGS=256" "MSAM2_TT_WGS=320" "MSAM2_TT_WGS=384" "MSAM2_TT_WGS=256 MSAM2_TT_MINK=32"; do
  echo "== $cfg"; env $cfg timeout -k 10 200 python tools/gemm_tt_bench.py 2>&1 | grep -v amdgpu | tail -1
done
echo "== WGS=256 all lines"; MSAM2_TT_WGS=256 timeout -k 10 200 python tools/gemm_tt_bench.py 2>&1 | grep -v amdgpu
