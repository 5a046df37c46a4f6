#!/bin/bash
# round 5, final kernels: every profile set of the round in one GPU session (run through gpurun)
R=$GRAFT_REPO_ROOT
cd $R
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3) > gpurun_out/r05_gputest_tail.txt
bash $R/profiles/collect_profiles.sh r05 c2 c3 c4 c5 c6 k6 c7 c8 c9
cd $R
timeout 600 python scripts/decompose_step.py --out gpurun_out/r05_step_decomposition.json > gpurun_out/r05_decomp.log 2>&1
for dim in 2 3; do timeout 300 python scripts/bench_generate.py --dim $dim; done > gpurun_out/r05_generate.jsonl 2> gpurun_out/r05_generate.err
rm -f gpurun_out/r05_sweep.jsonl
for c in c2 c3; do timeout 600 python bench.py --config $c --steps 50 --warmup 5 --repeats 1 --configs none --no-cpu-baseline --no-variants --no-verify --sweep --sweep-out gpurun_out/r05_sweep.jsonl > /dev/null 2> gpurun_out/r05_sweep_$c.err; done
for c in c6 c7 c8 c9; do
  bash profiles/collect_sq.sh r05 $c inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES
done
bash profiles/collect_sq.sh r05 c6 time SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
timeout 300 python scripts/calibrate_bw.py --out gpurun_out/r05_bw_calibration.json > /dev/null 2>&1
timeout 300 python scripts/calibrate_bw.py --store-shapes --out gpurun_out/r05_bw_store_shapes.json > /dev/null 2>&1
timeout 900 python scripts/time_shapes.py --out gpurun_out/r05_shapes.jsonl > /dev/null 2> gpurun_out/r05_shapes.err
timeout 900 python scripts/stress_rolling.py 200 gpurun_out/r05_stress_rolling.json > gpurun_out/r05_stress_rolling.log 2>&1
timeout 900 python scripts/stress_parity.py 1500 gpurun_out/r05_stress_parity.json > gpurun_out/r05_stress_parity.log 2>&1
SECONDS=0
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_default_bench.json 2> gpurun_out/r05_default_bench.err
echo "default bench: $SECONDS s" >> gpurun_out/r05_gputest_tail.txt
cat gpurun_out/r05_gputest_tail.txt
