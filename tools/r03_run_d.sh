#!/bin/bash
out=gpurun_out/r03d
mkdir -p $out
python tools/wstat_bench.py > $out/wstat.txt 2>&1; echo "wstat rc=$?"; cat $out/wstat.txt
for i in 1 2; do python -m pytest tests/test_autograd_gpu.py -m gpu -q -s -k "track_step" > $out/autograd_$i.log 2>&1; echo "autograd $i rc=$?"; grep -n "autograd loop vs\|fraction of\|^E  " $out/autograd_$i.log | cut -c1-700; done
python -m pytest tests/test_bf16_build_gpu.py -m gpu -q -s > $out/bf16.log 2>&1; echo "bf16 rc=$?"; tail -30 $out/bf16.log | cut -c1-300
