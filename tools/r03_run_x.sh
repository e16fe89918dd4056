#!/bin/bash
mkdir -p gpurun_out/r03x
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_graphs_gpu.py -m gpu -q -x -k "kv64 or graph" > gpurun_out/r03x/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r03x/tests.log
timeout -k 10 300 python tools/attn_ab.py build_ab/libold.so medical-sam2_amd/libmsam2_hip.so 2>&1 | grep -E "so:|memory"
