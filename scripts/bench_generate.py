#!/usr/bin/env python3
"""Throughput of device-side instance generation next to the C oracle on one host core.

    python scripts/bench_generate.py [--batch 65536] [--dim 2|3]
Prints one JSON line.  Reference for scale: generate.generate_blocks + calc_dependent take ~2.8 ms per
RAND n=10 instance in numpy, i.e. ~360 instances/s on one core.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import tap_net_amd as T            # noqa: E402
from tap_net_amd import generate   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--dim", type=int, default=2)
    ap.add_argument("--nodes", type=int, default=10)
    ap.add_argument("--reps", type=int, default=10)
    a = ap.parse_args()
    D, n, B = a.dim, a.nodes, a.batch
    generate.generate_instances(B, n, D, seed=1)                 # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in range(a.reps):
        st, dy = generate.generate_instances(B, n, D, seed=100 + r)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.reps
    # kernels alone: pack + precedence on a fixed accepted batch
    st, dy, blocks, pos = generate.generate_instances(B, n, D, seed=7, return_aux=True)
    cs = generate.initial_container(D, 7, 50)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for _ in range(a.reps):
        generate.pack_blocks(blocks, cs)
    e1.record()
    for _ in range(a.reps):
        generate.precedence_tensors(blocks, pos, cs)
    e2.record()
    torch.cuda.synchronize()
    import oracle_lib as O
    bl = blocks.cpu().numpy()
    m = min(B, 4000)
    t0 = time.perf_counter()
    for b in range(m):
        O.instance_from_blocks(bl[b], cs, 1)
    cpu = m / (time.perf_counter() - t0)
    print(json.dumps({
        "what": "RAND instance generation, D=%d n=%d, initial container %s" % (D, n, cs),
        "batch": B, "instances_per_s_end_to_end": B / dt,
        "pack_blocks_us": e0.elapsed_time(e1) * 1e3 / a.reps, "precedence_us": e1.elapsed_time(e2) * 1e3 / a.reps,
        "instances_per_s_kernels_only": B / ((e0.elapsed_time(e2)) * 1e-3 / a.reps),
        "cpu_oracle_instances_per_s_1core": cpu, "reference_numpy_instances_per_s": 360}))


if __name__ == "__main__":
    main()
