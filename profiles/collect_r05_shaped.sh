#!/bin/bash
# round 5, after the fused step got its compiled-in window shape: c2 / c3 profile sets again (the kernel names changed),
# the GPU tests the previous session did not run, the driver-like default line (run through gpurun)
R=$GRAFT_REPO_ROOT
cd $R
(timeout 1500 python -m pytest tests -m gpu -q -k "not (transition or golden or episode or stepper or mask or bench or c2 or smoke)" 2>&1 | tail -2) > gpurun_out/r05_gputest_rest_tail.txt
bash $R/profiles/collect_profiles.sh r05 c2 c3
cd $R
SECONDS=0
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_default_bench.json 2> gpurun_out/r05_default_bench.err
echo "default bench: $SECONDS s" >> gpurun_out/r05_gputest_rest_tail.txt
cat gpurun_out/r05_gputest_rest_tail.txt
