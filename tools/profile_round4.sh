#!/bin/bash
# Round-4 profile bundle (GPU box).  The headline library is the bf16 build (bench.py --dtype bf16, the default), so every pass loads it.
#  1. kernel-trace stats of the bench command (rocprofv3 --kernel-trace --stats) + the per-replay breakdown of tools/trace_step.py;
#  2. PMC passes, each in its own run and never with a trace domain: HBM bytes of the dominant kernel (memory cross-attention,
#     attn_kv64x2_kernel) for bench.py's roofline.traffic; SQ counters of the Hiera global attention kernel with the softmax reference
#     inside the MFMA (attn_g96x2_kernel, MREF) -- MFMA-busy, VALU instructions, waits.
tag=${1:-r04}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
export MSAM2_LIB_PATH=$PWD/medical-sam2_amd/libmsam2_hip_bf16.so
export MSAM2_BENCH_GEMM_TABLE=$out/gemm_table.json
python3 bench.py --steps 20 --warmup 3 > $out/bench_line.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-volume --no-bf16 > $out/bench_line_under_profiler.json 2> $out/bench_prof.err
rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-train --no-volume --no-bf16 --no-rooflines > /dev/null 2> $out/trace.err
python3 tools/trace_step.py $out/trace 5 > $out/step_breakdown.txt 2>&1
rm -rf $out/trace
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $out/kv64_$c -- python3 tools/one_attn.py 4 1 4096 16384 kv64 4 > /dev/null 2> $out/kv64_$c.err
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/g96_SQ -- python3 tools/one_attn.py 4 4 4096 4096 96 1 > /dev/null 2> $out/g96_SQ.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/g96_SQ_b -- python3 tools/one_attn.py 4 4 4096 4096 96 1 > /dev/null 2> $out/g96_SQ_b.err
find $out -name "*.csv" | sort | head -40
tail -c 600 $out/bench_line.json
