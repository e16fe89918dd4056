#!/bin/bash
# Round profile bundle (GPU box): kernel-trace stats of the bench step + PMC passes (each in its own run, never with a trace domain) of
# the dominant kernel and of the Hiera windowed attention.  Outputs under gpurun_out/prof_$1/
tag=${1:-r02}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train > $out/bench_line.json 2> $out/bench.err
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 tools/one_attn.py 4 1 4096 16384 kv64 4 > /dev/null 2> $out/pmc_$c.err
  rocprofv3 --pmc $c --output-format csv -d $out/win_pmc_$c -- python3 tools/one_win.py stage3 > /dev/null 2> $out/win_pmc_$c.err
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_SQ_a -- python3 tools/one_attn.py 4 1 4096 16384 kv64 4 > /dev/null 2> $out/pmc_SQ_a.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS --output-format csv -d $out/pmc_SQ_b -- python3 tools/one_attn.py 4 1 4096 16384 kv64 4 > /dev/null 2> $out/pmc_SQ_b.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/win_pmc_SQ_a -- python3 tools/one_win.py stage3 > /dev/null 2> $out/win_pmc_SQ_a.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS --output-format csv -d $out/win_pmc_SQ_b -- python3 tools/one_win.py stage3 > /dev/null 2> $out/win_pmc_SQ_b.err
find $out -name "*.csv" | head -40
