#!/bin/bash
# round 4, final kernels (one-wait stream waves, kernarg preload): every profile set of the round in one GPU session
R=$GRAFT_REPO_ROOT
cd $R
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3) > gpurun_out/r04_gputest_tail.txt
bash $R/profiles/collect_profiles.sh r04 c2 c3 c4 c5 c6 k6
cd $R
timeout 300 python scripts/decompose_step.py --out gpurun_out/r04_step_decomposition.json > gpurun_out/r04_decomp.log 2>&1
for dim in 2 3; do timeout 300 python scripts/bench_generate.py --dim $dim; done > gpurun_out/r04_generate.jsonl 2> gpurun_out/r04_generate.err
rm -f gpurun_out/r04_sweep.jsonl
for c in c2 c3 c6; do timeout 600 python bench.py --config $c --steps 50 --warmup 5 --repeats 1 --no-cpu-baseline --no-variants --no-verify --sweep --sweep-out gpurun_out/r04_sweep.jsonl > /dev/null 2> gpurun_out/r04_sweep_$c.err; done
for c in c2 c6; do
  bash profiles/collect_sq.sh r04 $c inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES
  bash profiles/collect_sq.sh r04 $c time SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
done
BENCH_FLAGS="--batch 1048576" bash profiles/collect_sq.sh r04 c2 inst1m SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES
BENCH_FLAGS="--batch 1048576" bash profiles/collect_sq.sh r04 c2 time1m SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
timeout 300 python scripts/calibrate_bw.py --out gpurun_out/r04_bw_calibration.json > /dev/null 2>&1
cat gpurun_out/r04_gputest_tail.txt
