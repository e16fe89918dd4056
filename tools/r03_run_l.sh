#!/bin/bash
# round 3 batch L: LDS-DMA issued from asm (no compiler vmcnt(0) before the tr-reads) against the builtin form
mkdir -p gpurun_out/r03l
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention or gelu or gemm_epilogue or ln_mlp or conv3x3 or layernorm" > gpurun_out/r03l/tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r03l/tests.log
timeout -k 10 300 python tools/attn_ab.py build_ab/libmsam2_hip_dmabuiltin.so medical-sam2_amd/libmsam2_hip.so > gpurun_out/r03l/ab.txt 2>&1; grep -v amdgpu.ids gpurun_out/r03l/ab.txt
MSAM2_G96_V1=1 timeout -k 10 300 python tools/attn_ab.py build_ab/libmsam2_hip_dmabuiltin.so medical-sam2_amd/libmsam2_hip.so > gpurun_out/r03l/ab_v1.txt 2>&1; grep -A1 "so:" gpurun_out/r03l/ab_v1.txt
