#!/bin/bash
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03am/train -- python3 $GRAFT_REPO_ROOT/tools/train_full_bench.py > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r03am/train -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/r03am/train_kernel_stats.csv
find $GRAFT_REPO_ROOT/gpurun_out/r03am/train -name "*kernel_trace.csv" -delete
python3 - <<'PY'
import csv, os
rows=list(csv.DictReader(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r03am/train_kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("per iteration under profiler ms", tot/13/1e6)
for r in rows[:45]:
    print(r['Name'][:95], r['Calls'], "avg", round(float(r['AverageNs'])/1e3,1), "ms/iter", round(float(r['TotalDurationNs'])/13/1e6,3))
PY
