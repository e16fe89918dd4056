#!/bin/bash
# Where the D = 96 global attention kernel's time goes: the same kernel with pieces removed (diagnostic builds -DMSAM2_G96_PROBE=n in
# build_ab/, see attn_g96_kernel): 2 no softmax, 3 + fragments read once, 4 + no DMA / barrier (MFMAs only).
# Build the diagnostic libraries first (here, on the CPU box; they travel with the snapshot):
#   for n in 2 3 4; do make -C medical-sam2_amd/csrc -j6 EXTRA=-DMSAM2_G96_PROBE=$n OUT=$PWD/build_ab/libprobe$n.so OBJDIR=$PWD/build_ab/obj_probe$n; done
#   make -C medical-sam2_amd/csrc -j6 EXTRA=-DMSAM2_DMA_BUILTIN OUT=$PWD/build_ab/libmsam2_hip_dmabuiltin.so OBJDIR=$PWD/build_ab/obj_dmab
# (build_ab/ is git-ignored; MSAM2_G96_X2=1 selects the 4-wave x 64-query shape.)
mkdir -p gpurun_out/g96probe
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention_vs_oracle" > gpurun_out/g96probe/tests.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/g96probe/tests.log
for lib in medical-sam2_amd/libmsam2_hip.so build_ab/libprobe2.so build_ab/libprobe3.so build_ab/libprobe4.so; do
  echo "== $lib"
  timeout -k 10 120 python tools/attn_ab.py $lib 2>&1 | grep global
done
echo "== old kernel"
MSAM2_G96_V1=1 timeout -k 10 120 python tools/attn_ab.py medical-sam2_amd/libmsam2_hip.so 2>&1 | grep global
