#!/bin/bash
out=gpurun_out/r03f
mkdir -p $out
python -m pytest tests/ -m gpu -q --deselect tests/test_bf16_build_gpu.py -k "not 512_slices" > $out/tests_all.log 2>&1; echo "tests rc=$?"; tail -8 $out/tests_all.log | cut -c1-300
bash tools/profile_round3.sh r03 > $out/profile.log 2>&1; echo "profile rc=$?"; tail -3 $out/profile.log
