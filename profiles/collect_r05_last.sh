#!/bin/bash
# round 5, last kernel state (3D windows on 8 stream waves per workgroup, windows without the fp32 expansion, c4's shape
# and width compiled in): the full GPU suite, smoke, c4's profile set again, the driver-like default line (run through
# gpurun; the c3 / c5 sets were collected by the previous version of this script, on the same k_transition / rolling kernels)
R=$GRAFT_REPO_ROOT
cd $R
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -2) > gpurun_out/r05_gputest_tail.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> gpurun_out/r05_gputest_tail.txt
bash $R/profiles/collect_profiles.sh r05 ${COLLECT_CONFIGS:-c4}
cd $R
SECONDS=0
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_default_bench.json 2> gpurun_out/r05_default_bench.err
echo "default bench: $SECONDS s" >> gpurun_out/r05_gputest_tail.txt
cat gpurun_out/r05_gputest_tail.txt
