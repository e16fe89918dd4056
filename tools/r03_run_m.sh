#!/bin/bash
# round 3 batch M: D = 96 global attention, 8 waves x 32 queries vs 4 waves x 64 queries, asm DMA vs builtin
mkdir -p gpurun_out/r03m
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention_vs_oracle" > gpurun_out/r03m/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r03m/tests.log
MSAM2_G96_X2=1 timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention_vs_oracle" > gpurun_out/r03m/tests_x2.log 2>&1
echo "tests x2 rc=$?"; tail -3 gpurun_out/r03m/tests_x2.log
echo "== 8 waves x 32 queries"
timeout -k 10 300 python tools/attn_ab.py build_ab/libmsam2_hip_dmabuiltin.so medical-sam2_amd/libmsam2_hip.so > gpurun_out/r03m/ab.txt 2>&1; grep -A1 "so:" gpurun_out/r03m/ab.txt
echo "== 4 waves x 64 queries"
MSAM2_G96_X2=1 timeout -k 10 300 python tools/attn_ab.py build_ab/libmsam2_hip_dmabuiltin.so medical-sam2_amd/libmsam2_hip.so > gpurun_out/r03m/ab_x2.txt 2>&1; grep -A1 "so:" gpurun_out/r03m/ab_x2.txt
