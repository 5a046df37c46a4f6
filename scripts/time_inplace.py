#!/usr/bin/env python3
"""What the fp32 `dynamic` tensor costs a decoding step, three ways: the default step object (two alternating tensors, all
3n rows written every step), ONE tensor updated in place (tap_stepper_buffers.dyn[0] == dyn[1]: 3 rows per step), no fp32
tensor at all (dyn = NULL).  A recorded tour through rollout.run_episode, captured in one hipGraph per mode, at BASELINE's
shapes.  Run on the GPU box:  python scripts/time_inplace.py [--out profiles/r06_inplace.jsonl]"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tap_net_amd as T
from tap_net_amd import synth, pack

SHAPES = [("c2", 2, [5, 50], 10, 8192, "C+P+S-lb-soft", "LB_GREEDY"), ("c3", 3, [5, 5, 50], 10, 4096, "C+P+S-lb-soft", "LB_GREEDY"),
          ("c4", 2, [7, 100], 20, 8192, "C+P+S-mcs-soft", "MACS"), ("c6", 3, [5, 5, 50], 10, 4096, "C+P+S-mcs-soft", "MACS"),
          ("c2x8", 2, [5, 50], 10, 65536, "C+P+S-lb-soft", "LB_GREEDY")]


def timed(st, dy, tape, cs, reward, strategy, n, steps=200, **kw):
    dev = st.device
    B = st.shape[0]
    env = T.BatchedContainer(B, cs, n, reward, "diff", packing_strategy=strategy, device=dev)
    sp = pack.EpisodeStepper(st, dy, env, steps=n, **kw)
    pol = T.TapePolicy(tape)
    run = lambda: T.run_episode(st, dy, pol, cs[0], cs[-1], reward_type=reward, packing_strategy=strategy, stepper=sp)  # noqa: E731
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        run(); run()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rec = run()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize(dev)
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    sp.check()
    return best / (steps * n) * 1e6, rec["reward"].clone(), (sp.dynamic.clone() if sp.dynamic is not None else None)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--shapes", default=None)
    ap.add_argument("--only", default=None, help="two | one | none: time one mode only (kernel traces)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    pack.set_binary_check('trust')
    for name, D, cs, n, B, reward, strategy in SHAPES:
        if args.shapes and name not in args.shapes.split(","):
            continue
        static, dynamic = synth.rand_instances(B, n, D, seed=7)
        tape = synth.random_feasible_tape(static, dynamic, n, seed=8).to(dev)
        st, dy = static.to(dev), dynamic.to(dev)
        if args.only:
            kw = dict(two={}, one=dict(inplace_dynamic=True), none=dict(expand_dynamic=False))[args.only]
            print(name, args.only, "%.3f us per step" % timed(st, dy, tape, cs, reward, strategy, n, **kw)[0], flush=True)
            continue
        two, r0, d0 = timed(st, dy, tape, cs, reward, strategy, n)
        one, r1, d1 = timed(st, dy, tape, cs, reward, strategy, n, inplace_dynamic=True)
        none, r2, _ = timed(st, dy, tape, cs, reward, strategy, n, expand_dynamic=False)
        same = bool(torch.equal(r0.nan_to_num(), r1.nan_to_num()) and torch.equal(r0.nan_to_num(), r2.nan_to_num()) and torch.equal(d0, d1))
        rec = dict(shape=name, B=B, n=n, container=cs, strategy=strategy, us_per_step=dict(two_buffers=two, in_place=one, no_fp32=none),
                   env_steps_per_s=dict(two_buffers=B / two * 1e6, in_place=B / one * 1e6, no_fp32=B / none * 1e6), identical=same)
        print(json.dumps(rec), flush=True)
        if args.out:
            with open(args.out, "a") as f:
                f.write(json.dumps(rec) + "\n")


if __name__ == "__main__":
    main()
