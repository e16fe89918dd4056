#!/bin/bash
for lib in medical-sam2_amd/libmsam2_hip.so build_ab/libil.so build_ab/libilprio.so build_ab/libprio.so; do
  echo "== $lib"
  timeout -k 10 120 python tools/attn_ab.py $lib 2>&1 | grep global
done
