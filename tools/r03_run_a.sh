#!/bin/bash
# round-3 GPU batch A: new tests, the bench line (N=1), the 2-rank launch rehearsal, the config-3 seed scan
out=gpurun_out/r03a
mkdir -p $out
python -m pytest tests/test_rccl_gpu.py tests/test_backward_gpu.py tests/test_backward_encoder_gpu.py tests/test_e2e_gpu.py -m gpu -x -q \
  -k "rccl or dropout or drift or chain or adam or train_step_2d" > $out/tests.log 2>&1; echo "tests rc=$?" | tee -a $out/tests.log
tail -5 $out/tests.log
python bench.py > $out/bench_n1.json 2> $out/bench_n1.err; echo "bench rc=$?"
python bench.py --gpus 2 --steps 5 --warmup 1 --slices 64 > $out/bench_n2_rehearsal.json 2> $out/bench_n2.err; echo "bench2 rc=$?"
python bench.py --mode volume --slices 64 --steps 2 > $out/bench_volume64.json 2> $out/bench_volume64.err; echo "benchvol rc=$?"
python tools/config3_seed_scan.py 64 0,1,2,3,4,5 0,1 > $out/seed_scan.txt 2>&1; echo "scan rc=$?"
cat $out/seed_scan.txt
