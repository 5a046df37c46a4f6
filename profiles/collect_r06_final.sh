#!/bin/bash
# round 6, final kernels: every profile set of the round in one GPU session (run through gpurun)
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3) > gpurun_out/r06_gputest_tail.txt
bash $R/profiles/collect_profiles.sh r06 c2 c3 c4 c5 c6 k6 c7 c8 c9
cd $R
timeout 600 python scripts/decompose_step.py --out gpurun_out/r06_step_decomposition.json > gpurun_out/r06_decomp.log 2>&1 < /dev/null
rm -f gpurun_out/r06_sweep.jsonl
for c in c2 c3; do timeout 600 python bench.py --config $c --steps 50 --warmup 5 --repeats 1 --configs none --no-cpu-baseline --no-variants --no-verify --sweep --sweep-out gpurun_out/r06_sweep.jsonl > /dev/null 2> gpurun_out/r06_sweep_$c.err < /dev/null; done
bash profiles/collect_sq.sh r06 c2 inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES
bash profiles/collect_sq.sh r06 c5 inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES
rm -f gpurun_out/r06_inplace.jsonl
timeout 600 python scripts/time_inplace.py --out gpurun_out/r06_inplace.jsonl > /dev/null 2> gpurun_out/r06_inplace.err < /dev/null
timeout 900 python scripts/time_shapes.py --out gpurun_out/r06_shapes.jsonl > /dev/null 2> gpurun_out/r06_shapes.err < /dev/null
timeout 900 python scripts/stress_rolling.py 200 gpurun_out/r06_stress_rolling.json > gpurun_out/r06_stress_rolling.log 2>&1 < /dev/null
timeout 1200 python scripts/stress_parity.py 1500 gpurun_out/r06_stress_parity.json > gpurun_out/r06_stress_parity.log 2>&1 < /dev/null
timeout 1200 python scripts/stress_masks.py 400 gpurun_out/r06_stress_masks.json > gpurun_out/r06_stress_masks.log 2>&1 < /dev/null
SECONDS=0
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_default_bench.json 2> gpurun_out/r06_default_bench.err < /dev/null
echo "default bench: $SECONDS s" >> gpurun_out/r06_gputest_tail.txt
cat gpurun_out/r06_gputest_tail.txt
