#!/bin/bash
# round 4, second GPU session: the remaining configs' profile sets, the generator figures, the batch sweeps and the SQ
# instruction counters (own --pmc passes, kernel trace only) at the BASELINE batch and at B = 1 M (c2 shape)
R=$GRAFT_REPO_ROOT
bash $R/profiles/collect_profiles.sh r04 c4 c5 c6 k6
cd $R
for dim in 2 3; do timeout 300 python scripts/bench_generate.py --dim $dim; done > gpurun_out/r04_generate.jsonl 2> gpurun_out/r04_generate.err
for c in c2 c3 c6; do timeout 600 python bench.py --config $c --steps 50 --warmup 5 --repeats 1 --no-cpu-baseline --no-variants --no-verify --sweep --sweep-out gpurun_out/r04_sweep.jsonl > /dev/null 2> gpurun_out/r04_sweep_$c.err; done
for c in c2 c6; do
  bash profiles/collect_sq.sh r04 $c inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES
  bash profiles/collect_sq.sh r04 $c time SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
done
BENCH_FLAGS="--batch 1048576" bash profiles/collect_sq.sh r04 c2 inst1m SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES
BENCH_FLAGS="--batch 1048576" bash profiles/collect_sq.sh r04 c2 time1m SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY
