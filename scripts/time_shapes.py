#!/usr/bin/env python3
"""Timings of the shape paths outside the BASELINE configs (round 3: two-word bit shadow, rolling windows over
65 .. 128 blocks, fused wide-MACS step), each next to the path the same shape took before.

    python scripts/time_shapes.py [--out profiles/r03_shapes.jsonl]

Whole episodes through the public host API (run_episode / run_rolling_episode with a tape policy), timed with
events on the launch stream after a warm-up episode; `us_per_step` = episode time / decoding steps, eager launches
(host launch overhead included: these are small-batch-insensitive comparisons, not roofline figures).
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tap_net_amd as T                      # noqa: E402
from tap_net_amd import synth, generate      # noqa: E402

DEV = "cuda:0"


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return best


def episode_case(name, B, n, D, cs, reward, strategy, variants):
    static, dynamic = synth.rand_instances(B, n, D, seed=5)
    tape = synth.random_feasible_tape(static, dynamic, n, seed=6).to(DEV)
    st, dy = static.to(DEV), dynamic.to(DEV)
    out = []
    ref = None
    for label, kw in variants:
        kw = dict(kw)
        if kw.pop("reuse", False):                        # what a trainer does: ONE container batch + step object per run
            env = T.BatchedContainer(B, cs, n, reward, "diff", packing_strategy=strategy, device=DEV)
            kw["stepper"] = T.pack.EpisodeStepper(st, dy, env, "bot", True, steps=n)
            if D == 3:
                kw["container_length"] = cs[1]

        def run(kw=kw):
            return T.run_episode(st, dy, T.TapePolicy(tape), cs[0], cs[-1], reward_type=reward,
                                 packing_strategy=strategy, **kw)
        r = run()
        if ref is None:
            ref = r["reward"]
        same = bool(torch.equal(torch.nan_to_num(r["reward"]), torch.nan_to_num(ref)))
        ms = timed(run)
        out.append(dict(case=name, variant=label, B=B, n=n, D=D, container=cs, us_per_step=1e3 * ms / n,
                        env_steps_per_s=B * n / (ms * 1e-3), same_rewards_as_first=same))
    return out


def rolling_case(name, B, N, child, D, init):
    static, dynamic, blocks, positions = generate.generate_instances(B, min(N, 64), D, init[0], init[-1], 1, (1, 5), seed=3,
                                                                     device=DEV, return_aux=True)
    if N > 64:                                            # the generator stops at 64 blocks: random blocks, packed soft
        g = torch.Generator(device=DEV)
        g.manual_seed(1)
        blocks = torch.randint(1, 5, (B, N, D), device=DEV, generator=g, dtype=torch.int32)
        positions, _, _ = generate.pack_blocks(blocks, init, 'C+P+S-lb-soft')
    H = min(4 * N + 10, 4000)
    gen = torch.Generator(device=DEV)

    def policy(step, static, dynamic, current_mask, **_):
        return torch.multinomial(current_mask, 1, generator=gen).squeeze(1)

    out = []
    ref = None
    for label, fused in (("fused", True), ("two launches", False)):
        def run():
            gen.manual_seed(7)
            return T.run_rolling_episode(blocks, positions, init, policy, 5, H, child_graph_size=child, fused=fused)
        r = run()
        if ref is None:
            ref = r["reward"]
        ms = timed(run, reps=3)
        out.append(dict(case=name, variant=label, B=B, N=N, child=child, D=D, us_per_step=1e3 * ms / N,
                        env_steps_per_s=B * N / (ms * 1e-3), same_rewards_as_first=bool(torch.equal(r["reward"], ref))))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default=None, help="substrings (comma-separated) of the case names to run")
    a = ap.parse_args()
    rows = []
    global episode_case, rolling_case
    if a.only:                                            # skip the other cases without touching the list below
        ep0, ro0 = episode_case, rolling_case
        episode_case = lambda name, *x, **k: ep0(name, *x, **k) if any(o in name for o in a.only.split(',')) else []     # noqa: E731
        rolling_case = lambda name, *x, **k: ro0(name, *x, **k) if any(o in name for o in a.only.split(',')) else []     # noqa: E731
    bits_vs_copy = [("fused step, bit shadow", dict(fused=True)), ("fused step, fp32 copy", dict(fused=True, bits=False)),
                    ("two launches, bit shadow", dict(fused=False)), ("two launches, fp32 copy", dict(fused=False, bits=False))]
    rows += episode_case("2D n=30 W=5 LB_GREEDY (90 rows: two-word shadow)", 8192, 30, 2, [5, 150], "C+P+S-lb-soft", "LB_GREEDY", bits_vs_copy)
    rows += episode_case("3D n=30 5x5 LB_GREEDY (90 rows, 180 columns)", 2048, 30, 3, [5, 5, 150], "C+P+S-lb-soft", "LB_GREEDY", bits_vs_copy)
    rows += episode_case("2D n=20 W=7 LB_GREEDY (60 rows: one-word shadow, for scale)", 8192, 20, 2, [7, 100], "C+P+S-lb-soft", "LB_GREEDY", bits_vs_copy)
    rows += episode_case("2D n=12 W=20 MACS (one wavefront per container since the end of round 4)", 8192, 12, 2, [20, 80], "C+P+S-mcs-soft", "MACS", bits_vs_copy[:1] + bits_vs_copy[2:3])
    rows += episode_case("2D n=10 W=40 MACS (one wavefront per container)", 4096, 10, 2, [40, 60], "C+P+S-mcs-soft", "MACS", bits_vs_copy[:1] + bits_vs_copy[2:3])
    rows += rolling_case("3D rolling N=50 child=10 (one-word graphs, for scale)", 4096, 50, 10, 3, [7, 7, 250])
    rows += rolling_case("3D rolling N=100 child=10 (two-word graphs, one wavefront per instance)", 4096, 100, 10, 3, [7, 7, 500])
    rows += rolling_case("3D rolling N=128 child=10", 4096, 128, 10, 3, [7, 7, 600])
    rows += rolling_case("3D rolling N=130 child=10 (three-word graphs, one wavefront per instance; round 4: one thread)", 4096, 130, 10, 3, [7, 7, 600])
    rows += rolling_case("3D rolling N=200 child=10 (four-word graphs, one wavefront per instance)", 4096, 200, 10, 3, [7, 7, 900])
    rows += rolling_case("3D rolling N=256 child=10 (four-word graphs)", 4096, 256, 10, 3, [7, 7, 1100])
    rows += rolling_case("3D rolling N=300 child=10 (one thread per instance, 16-word masks)", 1024, 300, 10, 3, [7, 7, 1300])
    # round 4: the one-thread-per-container paths (correctness paths for unusual --container_width values), next to the
    # lane-per-cell kernels at the nearest shapes they cover
    one = bits_vs_copy[:1] + [("fused step, bit shadow, step object re-used across episodes", dict(fused=True, reuse=True))]
    rows += episode_case("3D n=10 8x8 LB_GREEDY (lane per cell, for scale)", 4096, 10, 3, [8, 8, 50], "C+P+S-lb-soft", "LB_GREEDY", one)
    rows += episode_case("3D n=10 10x10 LB_GREEDY (big.hip: one wavefront per container)", 4096, 10, 3, [10, 10, 50], "C+P+S-lb-soft", "LB_GREEDY", one)
    rows += episode_case("3D n=10 10x10 LB_GREEDY hard rewards (big.hip)", 4096, 10, 3, [10, 10, 50], "C+P+S-lb-hard", "LB_GREEDY", one)
    rows += episode_case("3D n=20 16x16 LB_GREEDY (big.hip)", 4096, 20, 3, [16, 16, 60], "C+P+S-lb-soft", "LB_GREEDY", one)
    rows += episode_case("3D n=10 8x8 MACS (lane per cell, for scale)", 4096, 10, 3, [8, 8, 50], "C+P+S-mcs-soft", "MACS", one)
    rows += episode_case("3D n=10 10x10 MACS (macs3_big.hip: one wavefront per container)", 4096, 10, 3, [10, 10, 50], "C+P+S-mcs-soft", "MACS", one)
    rows += episode_case("2D n=10 W=64 MACS (one wavefront per container)", 4096, 10, 2, [64, 50], "C+P+S-mcs-soft", "MACS", one)
    rows += episode_case("2D n=10 W=100 MACS (macs_big.hip: one wavefront per container)", 4096, 10, 2, [100, 50], "C+P+S-mcs-soft", "MACS", one)
    rows += episode_case("2D n=10 W=100 LB_GREEDY (big.hip)", 4096, 10, 2, [100, 50], "C+P+S-lb-soft", "LB_GREEDY", one)
    for r in rows:
        print(json.dumps(r))
    if a.out:
        with open(a.out, "w") as f:
            for r in rows:
                f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main()
