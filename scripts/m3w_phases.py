#!/usr/bin/env python3
"""Cycles per phase of the wave-per-container MACS 3D placement (tap_macs3_wave.h, -DM3W_PROF scratch build):
    TAP_LIB_PATH=build_prof/libtapenv_m3wprof.so python scripts/m3w_phases.py
Phases: 0 level intervals (a), 1 block-adjacent spaces (b), 2 position table, 3 corner walks, 4 scores, 5 tie-break, 6 commit."""
import ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                          # noqa: E402
import tap_net_amd as T               # noqa: E402
lib = ctypes.CDLL(os.environ["TAP_LIB_PATH"])
NAMES = ["intervals(a)", "blocks(b)", "table", "walks", "scores", "tiebreak", "commit"]
for cs, n, B in (([10, 10, 50], 10, 4096), ([12, 12, 40], 20, 2048), ([20, 20, 30], 30, 1024)):
    rng = np.random.RandomState(1)
    blocks = torch.as_tensor(rng.randint(1, 5, size=(B, n, 3)).astype(np.int32), device="cuda:0")
    env = T.BatchedContainer(B, cs, n, "C+P+S-mcs-soft", "diff", packing_strategy="MACS", device="cuda:0")
    out = (ctypes.c_ulonglong * 16)()
    for rep in range(2):
        env.reset(); torch.cuda.synchronize(); lib.tap_m3w_prof_read(out, 1); t0 = time.perf_counter()
        for t in range(n):
            env.add_new_blocks(blocks[:, t].contiguous())
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    lib.tap_m3w_prof_read(out, 1)
    tot = float(sum(out[:7]))
    print(json.dumps(dict(container=cs, n=n, B=B, us_per_step=dt / n * 1e6,
                          cycles_per_placement=tot / (B * n),
                          share={NAMES[k]: round(out[k] / tot, 3) for k in range(7)},
                          per_placement=dict(spaces=out[8] / max(out[12], 1), walks=out[9] / max(out[12], 1), corner_hits=out[10] / max(out[12], 1),
                                             searches_that_found=out[11] / max(out[12], 1)))))
