#!/usr/bin/env python3
"""Eager decoding steps on the wave-per-container kernels (big.hip, macs_big.hip, macs3_big.hip), for a kernel trace:
    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_wave -o wave -- python $OLDPWD/scripts/time_wave_kernels.py
prints the eager microseconds per step; the per-kernel durations are in the trace's kernel_stats."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                          # noqa: E402
import tap_net_amd as T               # noqa: E402
DEV = "cuda:0"
CASES = [([10, 10, 50], 10, 4096, "C+P+S-lb-soft", "LB_GREEDY"), ([10, 10, 50], 10, 4096, "C+P+S-lb-hard", "LB_GREEDY"),
         ([10, 10, 50], 10, 4096, "C+P+S-mcs-soft", "MACS"), ([20, 20, 30], 30, 1024, "C+P+S-mcs-soft", "MACS"),
         ([32, 50], 10, 4096, "C+P+S-mcs-soft", "MACS"), ([100, 50], 10, 4096, "C+P+S-mcs-soft", "MACS"), ([100, 50], 10, 4096, "C+P+S-lb-soft", "LB_GREEDY")]
for cs, n, B, reward, strat in CASES:
    rng = np.random.RandomState(1)
    blocks = torch.as_tensor(rng.randint(1, 5, size=(B, n, len(cs))).astype(np.int32), device=DEV)
    env = T.BatchedContainer(B, cs, n, reward, "diff", packing_strategy=strat, device=DEV)
    for rep in range(3):
        env.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in range(n):
            env.add_new_blocks(blocks[:, t].contiguous())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(cs, strat, reward, "B", B, "%.1f us per eager step" % (dt / n * 1e6))
