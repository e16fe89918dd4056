#!/bin/bash
# round 3 batch Z: full fp16 suite (incl. the 512-slice test), then the bf16 suite
mkdir -p gpurun_out/r03z
python -m pytest tests/ -m gpu -q --deselect tests/test_bf16_build_gpu.py > gpurun_out/r03z/tests_all.log 2>&1; echo "fp16 rc=$?"; tail -4 gpurun_out/r03z/tests_all.log
python -m pytest tests/test_bf16_build_gpu.py -m gpu -q -s > gpurun_out/r03z/bf16.log 2>&1; echo "bf16 rc=$?"; tail -6 gpurun_out/r03z/bf16.log | cut -c1-300
