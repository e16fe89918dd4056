#!/bin/bash
# GEMM family evidence (GPU box): per-shape census of the bench step + PMC passes (each in its own run) on the two stage-3 shapes that
# carry half of the GEMM time (16384 x 1536 x 384 = fc1, 16384 x 384 x 1536 = fc2).  Outputs under gpurun_out/prof_gemm_$1/
tag=${1:-r03}
out=gpurun_out/prof_gemm_$tag
mkdir -p $out
export TMPDIR=/tmp
python3 tools/gemm_census.py > $out/gemm_census.txt 2> $out/census.err
for shape in "16384 1536 384" "16384 384 1536"; do
  s=$(echo $shape | tr ' ' 'x')
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/${s}_$c -- python3 tools/one_gemm.py $shape > /dev/null 2> $out/${s}_$c.err
  done
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/${s}_TCC -- python3 tools/one_gemm.py $shape > /dev/null 2> $out/${s}_TCC.err
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/${s}_SQ_a -- python3 tools/one_gemm.py $shape > /dev/null 2> $out/${s}_SQ_a.err
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS --output-format csv -d $out/${s}_SQ_b -- python3 tools/one_gemm.py $shape > /dev/null 2> $out/${s}_SQ_b.err
done
find $out -name "*counter_collection.csv" | sort
