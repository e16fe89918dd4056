#!/bin/bash
# end-of-round profile of the bench step (kernel trace stats only) + the dominant kernel's traffic counters
out=gpurun_out/prof_r03e
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-volume --no-bf16 > $out/bench_line.json 2> $out/bench.err
f=$(find $out/bench -name "*kernel_stats.csv" | head -1); cp $f $out/bench_kernel_stats.csv
find $out/bench -name "*kernel_trace.csv" -delete
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 tools/one_attn.py 4 1 4096 16384 kv64 4 > /dev/null 2> $out/pmc_$c.err
  f=$(find $out/pmc_$c -name "*counter_collection.csv" | head -1); cp $f $out/attnkv64_pmc_$c.csv
done
cut -c1-300 $out/bench_line.json
head -32 $out/bench_kernel_stats.csv | cut -c1-130
