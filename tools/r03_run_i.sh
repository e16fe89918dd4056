#!/bin/bash
out=gpurun_out/r03i
mkdir -p $out
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-train --no-volume --no-bf16 --no-rooflines 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('batched kproj   ', d['value'], d['ms_per_step'])"
  MSAM2_NO_BATCHED_KPROJ=1 python bench.py --no-cpu-baseline --no-train --no-volume --no-bf16 --no-rooflines 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('per-layer kproj ', d['value'], d['ms_per_step'])"
done
python -m pytest tests/test_backward_gpu.py -m gpu -q -k "dropout" > $out/dropout.log 2>&1; echo "dropout rc=$?"; tail -12 $out/dropout.log | cut -c1-300
python tools/volume_bench.py 64 1 2>&1 | tail -6 | cut -c1-200
MSAM2_NO_BATCHED_KPROJ=1 python tools/volume_bench.py 64 1 2>&1 | tail -6 | cut -c1-200
