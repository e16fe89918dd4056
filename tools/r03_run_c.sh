#!/bin/bash
out=gpurun_out/r03c
mkdir -p $out
python tools/wstat_bench.py > $out/wstat.txt 2>&1; echo "wstat rc=$?"; cat $out/wstat.txt
python -m pytest tests/test_autograd_gpu.py tests/test_backward_encoder_gpu.py tests/test_volume_ranks_gpu.py tests/test_rccl_gpu.py tests/test_kernels_gpu.py tests/test_modules_gpu.py -m gpu -q > $out/tests1.log 2>&1; echo "tests1 rc=$?"; tail -12 $out/tests1.log
python -m pytest tests/test_e2e_gpu.py -m gpu -q -k "not 512_slices" > $out/e2e.log 2>&1; echo "e2e rc=$?"; tail -6 $out/e2e.log
python -m pytest tests/test_bf16_build_gpu.py -m gpu -q -s > $out/bf16.log 2>&1; echo "bf16 rc=$?"; tail -30 $out/bf16.log | cut -c1-300
