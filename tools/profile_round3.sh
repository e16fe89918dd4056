#!/bin/bash
# Round-3 profile bundle (GPU box): kernel-trace stats of the bench step; PMC passes (each in its own run, never with a trace domain) of the
# dominant kernel (memory cross-attention), the Hiera windowed attention, and the qkv GEMM shape on the W-stationary and the tiled kernel.
tag=${1:-r03}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-volume --no-bf16 > $out/bench_line.json 2> $out/bench.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 tools/one_attn.py 4 1 4096 16384 kv64 4 > /dev/null 2> $out/pmc_$c.err
  rocprofv3 --pmc $c --output-format csv -d $out/wstat_$c -- python3 tools/one_gemm.py 16384 1152 384 > /dev/null 2> $out/wstat_$c.err
done
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/wstat_TCC -- python3 tools/one_gemm.py 16384 1152 384 > /dev/null 2> $out/wstat_TCC.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/wstat_SQ_a -- python3 tools/one_gemm.py 16384 1152 384 > /dev/null 2> $out/wstat_SQ_a.err
export MSAM2_GEMM_WSTAT=0
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/tiled_$c -- python3 tools/one_gemm.py 16384 1152 384 > /dev/null 2> $out/tiled_$c.err
done
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/tiled_TCC -- python3 tools/one_gemm.py 16384 1152 384 > /dev/null 2> $out/tiled_TCC.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/tiled_SQ_a -- python3 tools/one_gemm.py 16384 1152 384 > /dev/null 2> $out/tiled_SQ_a.err
unset MSAM2_GEMM_WSTAT
find $out -name "*.csv" | sort | head -40
