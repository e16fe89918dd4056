#!/bin/bash
# round 3 batch P: bench line + full fp16 suite with the new global attention kernel, the polynomial GELU and GELU on the W-stationary kernel
mkdir -p gpurun_out/r03p
python bench.py --steps 20 --warmup 5 > gpurun_out/r03p/bench.json 2> gpurun_out/r03p/bench.err; echo "bench rc=$?"; cat gpurun_out/r03p/bench.json
MSAM2_GEMM_WSTAT=3 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train --no-volume --no-bf16 > gpurun_out/r03p/bench_wstat3.json 2> gpurun_out/r03p/bench_wstat3.err; echo "bench(wstat3) rc=$?"; cut -c1-400 gpurun_out/r03p/bench_wstat3.json
python -m pytest tests/ -m gpu -q --deselect tests/test_bf16_build_gpu.py -k "not 512_slices" > gpurun_out/r03p/tests_all.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r03p/tests_all.log
