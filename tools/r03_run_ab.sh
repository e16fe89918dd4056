#!/bin/bash
for v in "" "MSAM2_NO_ARENA=1" "MSAM2_NO_REFRESH=1" "MSAM2_NO_ARENA=1 MSAM2_NO_REFRESH=1" ""; do
  echo "== $v"; env $v timeout -k 10 300 python tools/train_full_bench.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-60
done
