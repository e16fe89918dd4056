#!/bin/bash
out=gpurun_out/r03h
mkdir -p $out
python bench.py > $out/bench_full.json 2> $out/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r03h/bench_full.json") if l.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"], "gemm", d["roofline_gemm"]["gemm_ms_per_step"], d["roofline_gemm"]["frac"], "train", d["train_iteration"].get("ms"), d["train_iteration_frozen_encoder"].get("ms"), "bf16", d["bf16"].get("value"), "vol", d["volume_3d"].get("slices_per_s"))
PY
MSAM2_NO_BATCHED_KPROJ=1 python bench.py --no-cpu-baseline --no-train --no-volume --no-bf16 --no-rooflines 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('no batched kproj', d['value'], d['ms_per_step'])"
python -m pytest tests/test_e2e_gpu.py tests/test_graphs_gpu.py tests/test_volume_ranks_gpu.py tests/test_modules_gpu.py -m gpu -q -k "not 512_slices" > $out/tests.log 2>&1; echo "tests rc=$?"; tail -4 $out/tests.log | cut -c1-300
python -m pytest tests/test_bf16_build_gpu.py -m gpu -q -s > $out/bf16.log 2>&1; echo "bf16 rc=$?"; tail -16 $out/bf16.log | cut -c1-300
python tools/copy_sites_train.py > $out/copy_sites.txt 2>&1; head -25 $out/copy_sites.txt | cut -c1-200
