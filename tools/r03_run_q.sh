#!/bin/bash
mkdir -p gpurun_out/r03q
MSAM2_E2E_REPORT=gpurun_out/r03q/report_poly.json timeout -k 10 300 python -m pytest tests/test_e2e_gpu.py -m gpu -q -k "chain_hiera_bplus_256 or chain_hiera_t or chain_hiera_s_256" > gpurun_out/r03q/poly.log 2>&1; tail -3 gpurun_out/r03q/poly.log
MSAM2_LIB_PATH=$PWD/build_ab/libgeluas.so MSAM2_E2E_REPORT=gpurun_out/r03q/report_as.json timeout -k 10 300 python -m pytest tests/test_e2e_gpu.py -m gpu -q -k "chain_hiera_bplus_256 or chain_hiera_t or chain_hiera_s_256" > gpurun_out/r03q/as.log 2>&1; tail -3 gpurun_out/r03q/as.log
python - <<'PY'
import json
for n in ("poly","as"):
    try:
        r=json.load(open(f"gpurun_out/r03q/report_{n}.json"))
        print(n, {k:v for k,v in r.items() if "_t" in k and k[0] in "bst" and "256" in k})
    except Exception as e: print(n, e)
PY
