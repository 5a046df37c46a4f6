#!/bin/bash
# round 5, after the MACS 3D kernels got their compile-time-sides instantiation (5 x 5): the full GPU suite, c6's profile
# set again, a MACS 3D parity sweep and the driver-like default line (run through gpurun)
R=$GRAFT_REPO_ROOT
cd $R
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3) > gpurun_out/r05_gputest_tail.txt
bash $R/profiles/collect_profiles.sh r05 c6
cd $R
bash profiles/collect_sq.sh r05 c6 inst SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES
STRESS_ONLY_MOD20=0,5,10,15 timeout 900 python scripts/stress_parity.py 1000 gpurun_out/r05_stress_macs3d.json > gpurun_out/r05_stress_macs3d.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/r05_gputest_tail.txt 2>&1
SECONDS=0
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_default_bench.json 2> gpurun_out/r05_default_bench.err
echo "default bench: $SECONDS s" >> gpurun_out/r05_gputest_tail.txt
cat gpurun_out/r05_gputest_tail.txt
tail -1 gpurun_out/r05_stress_macs3d.log | cut -c1-400
