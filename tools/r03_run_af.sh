#!/bin/bash
mkdir -p gpurun_out/r03af
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03af/bench.json 2> gpurun_out/r03af/bench.err ) 2>&1 | grep real; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03af/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline_gemm"]["frac"], d["roofline_gemm"]["gemm_ms_per_step"], {k:round(v["frac"],3) for k,v in d["roofline_hiera_attention"].items()})
print(d["train_iteration"]["ms"], d["train_iteration_frozen_encoder"]["ms"], d["bf16"]["value"], d["volume_3d"]["slices_per_s"], d["cpu_baseline"]["value"])
PY
