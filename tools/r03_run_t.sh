#!/bin/bash
mkdir -p gpurun_out/r03t
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03t/train -- python3 $GRAFT_REPO_ROOT/tools/train_full_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r03t/train.txt 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/r03t/train -name "*kernel_stats.csv" | head -1); cp $f $GRAFT_REPO_ROOT/gpurun_out/r03t/train_kernel_stats.csv; head -45 $f | cut -c1-150
find $GRAFT_REPO_ROOT/gpurun_out/r03t/train -name "*kernel_trace.csv" -delete
