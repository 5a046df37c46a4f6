#!/usr/bin/env python3
"""Throughput of the per-env seam S2 as the unchanged model.py drives it (model.py:294, 451-453, 509-510: one
tools.Container per env, add_new_block per env per decoding step, calc_ratio per env at the end) with the Containers
pooled on one BatchedContainer (tools.lockstep_containers: one launch per decoding step) and un-pooled (one one-env
launch + host sync per call).  GPU box:

    python scripts/time_lockstep.py [--batch 8192] [--unpooled-batch 512] >> profiles/r04_lockstep.jsonl
"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                          # noqa: E402
import tap_net_amd as T               # noqa: E402,F401
from tap_net_amd import tools         # noqa: E402


def loop(B, D, n, pooled, reps):
    cs = [5, 50] if D == 2 else [5, 5, 50]
    rng = np.random.RandomState(1)
    blocks = rng.randint(1, 5, size=(B, n, D)).astype(np.float32)
    prev = tools.lockstep_containers(pooled)
    try:
        best = None
        for _ in range(reps + 1):                                      # the first repetition warms up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            containers = [tools.Container(cs, n, "C+P+S-lb-soft", "diff", packing_strategy="LB_GREEDY", device="cuda:0") for _ in range(B)]
            for t in range(n):
                heightmaps = [containers[b].add_new_block(blocks[b, t], False) for b in range(B)]    # model.py:451-453
                torch.FloatTensor(np.array(heightmaps))                                              # model.py:454-465
            scores = [containers[b].calc_ratio() for b in range(B)]                                  # model.py:509-510
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None or _ == 1 else min(best, dt)
        return B * n / best, float(np.mean(scores))
    finally:
        tools.lockstep_containers(prev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--unpooled-batch", type=int, default=512)
    ap.add_argument("--nodes", type=int, default=10)
    a = ap.parse_args()
    for D in (2, 3):
        pooled, m1 = loop(a.batch, D, a.nodes, True, 3)
        solo, _ = loop(a.unpooled_batch, D, a.nodes, False, 1)
        print(json.dumps(dict(what="model.py's own per-env loop over tools.Container (S2 seam), %dD n=%d" % (D, a.nodes),
                              pooled_env_steps_per_s=pooled, pooled_batch=a.batch,
                              unpooled_env_steps_per_s=solo, unpooled_batch=a.unpooled_batch, mean_ratio=m1)))


if __name__ == "__main__":
    main()
