#!/bin/bash
mkdir -p gpurun_out/r03n
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attention_vs_oracle" > gpurun_out/r03n/tests.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r03n/tests.log
for i in 1 2; do
timeout -k 10 120 python tools/attn_ab.py medical-sam2_amd/libmsam2_hip.so 2>&1 | grep global
MSAM2_G96_X2=1 timeout -k 10 120 python tools/attn_ab.py medical-sam2_amd/libmsam2_hip.so 2>&1 | grep global
MSAM2_G96_V1=1 timeout -k 10 120 python tools/attn_ab.py medical-sam2_amd/libmsam2_hip.so 2>&1 | grep global
done
cd /tmp && export TMPDIR=/tmp
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03n/pmc_$tag -- python3 $GRAFT_REPO_ROOT/tools/one_attn.py 4 4 4096 4096 96 1 > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r03n/pmc_$tag -name "*counter_collection.csv" | head -1)
  echo "-- $c"; python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "g96" in r["Kernel_Name"] or "glds" in r["Kernel_Name"]:
        acc[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, "n=%d avg=%.4g" % (len(v), sum(v) / len(v)))
PY
done
