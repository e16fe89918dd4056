#!/bin/bash
# A/B of attn_kv64_kernel build variants (tools/variants/lib_kv64_<occ>_<kpf>.so) at the benchmark's shape; GPU only.
for lib in medical-sam2_amd/libmsam2_hip.so tools/variants/lib_kv64_*.so; do
  echo "== $lib"
  MSAM2_LIB_PATH=$PWD/$lib python tools/attn_kv64_bench.py 2>/dev/null | grep "round [12]" | grep -v "splits  8"
done
