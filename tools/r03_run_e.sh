#!/bin/bash
out=gpurun_out/r03e
mkdir -p $out
python bench.py --no-cpu-baseline --no-train --no-volume --no-bf16 > $out/bench_wstat.json 2> $out/bench.err; echo "bench rc=$?"
MSAM2_GEMM_WSTAT=0 python bench.py --no-cpu-baseline --no-train --no-volume --no-bf16 --no-rooflines > $out/bench_nowstat.json 2>> $out/bench.err; echo "bench0 rc=$?"
python - <<'PY'
import json
for f in ("bench_wstat.json","bench_nowstat.json"):
    d=json.loads([l for l in open("gpurun_out/r03e/"+f) if l.startswith("{")][-1]); print(f, d["value"], d["ms_per_step"], d.get("roofline_gemm",{}).get("gemm_ms_per_step"), d.get("roofline_gemm",{}).get("frac"))
PY
python -m pytest tests/ -m gpu -q -x --deselect tests/test_bf16_build_gpu.py -k "not 512_slices" > $out/tests_all.log 2>&1; echo "tests rc=$?"; tail -8 $out/tests_all.log | cut -c1-300
