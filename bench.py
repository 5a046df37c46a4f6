#!/usr/bin/env python3
"""Benchmark of the batched Transport-and-Pack environment hot path on MI355X.

One "step" = one full pass of the hot path over one batch of synthetic instances:
    reset -> n x [ update_dynamic + update_mask (one launch) ; add_new_block (one launch) ]
          -> calc_ratio (one launch) [-> when N > 1, the (B,) reward vectors of 8 consecutive passes
             are all-gathered with one RCCL call]
i.e. what DRL.forward does around its policy network for one batch (model.py:294-515), with the
actions replayed from a pre-computed feasible tape.  value = env-steps/s = (placements in all
envs on all ranks) / wall time, inputs resident in HBM before the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5] [--batch B]
                    [--no-graph] [--no-cpu-baseline] [--sweep]

For N > 1 launch with torch.distributed.run (one rank per GPU, RCCL); rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import tap_net_amd as T                     # noqa: E402
from tap_net_amd import _lib, synth         # noqa: E402
from tap_net_amd import dist as tdist       # noqa: E402

HBM_PEAK_GBS = 8000.0    # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md (6.29 TB/s measured copy)

# nodes of one precedence window (rolling.py feeds the actor 10-node sub-graphs)
WINDOW = {"c5": 10}
# configs whose pass is rolling.validate's loop: N - 10 one-step windows re-cut from the precedence DAG
# after every placement (tap_rolling_window + tap_env_step_gather), then a full episode on the last one
ROLLING = {"c5"}

CONFIGS = {
    # name: (workload string, D, container, n, per-GPU batch, reward, strategy)
    "c2": ("2D RAND nodes=10 container_width=5 LB_GREEDY batch=8192 on 1xMI355X (BASELINE configs[1])",
           2, [5, 50], 10, 8192, "C+P+S-lb-soft", "LB_GREEDY"),
    "c3": ("3D RAND nodes=10 container_width=5x5 LB_GREEDY batch=4096 on 1xMI355X (BASELINE configs[2])",
           3, [5, 5, 50], 10, 4096, "C+P+S-lb-soft", "LB_GREEDY"),
    "c4": ("2D nodes=20 container_width=7 MACS batch=8192 on 1xMI355X (BASELINE configs[3], RAND-marginal blocks)",
           2, [7, 100], 20, 8192, "C+P+S-mcs-soft", "MACS"),
    "c5": ("3D nodes=50 (5 consecutive 10-node precedence windows) container_width=5x5 H=250 LB_GREEDY "
           "batch=8192 per GPU (BASELINE configs[4] shard)",
           3, [5, 5, 250], 50, 8192, "C+P+S-lb-soft", "LB_GREEDY"),
    "c6": ("3D RAND nodes=10 container_width=5x5 MACS batch=4096 on 1xMI355X (not a BASELINE config: SURVEY 8(f) f3)",
           3, [5, 5, 50], 10, 4096, "C+P+S-mcs-soft", "MACS"),
}


def algorithmic_bytes(D, cs, n):
    """SURVEY.md 8(d) per-env-step figures (int32 state as the neutral unit)."""
    cells = int(np.prod(cs[:-1]))
    flen = cs[0] - 1 if D == 2 else 2 * cells
    env = (cells * 4 + D * 4 + 16) + (cells * 4 + 16 + D * 4 + 4 + flen * 4)
    R = 2 if D == 2 else 6
    nR = n * R
    mask = (3 * n * nR * 4 + nR * 4 + 8) + (3 * n * nR * 4 + 2 * nR * 4)
    return env, mask


class HotPath(object):
    """Pre-allocated buffers + direct C-ABI calls for one rank's share of the batch.

    An episode is `windows` consecutive precedence windows of `nw` nodes each over ONE long-lived
    container per env (windows = 1 except for the rolling-sized config c5)."""

    def __init__(self, cfg, B, start, device, seed=12345, fused=True, window=None, bits=True, instances=None):
        _, D, cs, n, _, reward, strategy = cfg
        self.D, self.cs, self.n, self.B, self.device = D, cs, n, B, device
        self.nw = window or n
        assert n % self.nw == 0
        self.windows = n // self.nw
        f32 = dict(dtype=torch.float32, device=device)
        self.instances = instances
        self.static, self.dynamic0, self.tape, self.cs0, self.bits0 = [], [], [], [], []
        for w in range(self.windows):
            if instances == "generate":                             # RAND instances of the device-side generator (f1)
                static, dynamic = synth.device_rand_instances(B, self.nw, D, seed=seed + 100 * w, start=start, device=device)
                static, dynamic = static.cpu(), dynamic.cpu()
            elif instances is not None:                             # real instances from a committed fixture, tiled
                static, dynamic = synth.tiled_instances(instances[0], instances[1], B, start=start)
            else:
                static, dynamic = synth.rand_instances(B, self.nw, D, seed=seed + 100 * w, start=start)
            tape = synth.random_feasible_tape(static, dynamic, self.nw, seed=seed + 100 * w + 1, start=start)
            self.static.append(static.to(device))
            self.dynamic0.append(dynamic.to(device))
            self.tape.append(tape.t().contiguous().to(device))  # (nw, B): one contiguous ptr row per step
            self.cs0.append(T.pack.dynamic_colsum(self.dynamic0[-1], self.nw).clone())
            # the instance's bit shadow, like its column sums, is dataset-time data (built once per
            # instance, not per pass); only valid for 0/1 tensors of a supported shape
            ok = bits and T.pack.bits_supported(self.dynamic0[-1].shape[1], self.dynamic0[-1].shape[2])
            shadow, bad = T.pack.dynamic_bits(self.dynamic0[-1]) if ok else (None, None)
            self.bits0.append(shadow if ok and int(bad.item()) == 0 else None)
        self.bits = all(b is not None for b in self.bits0)
        self.R = self.static[0].shape[2] // self.nw
        self.nR, self.rows = self.static[0].shape[2], self.dynamic0[0].shape[1]
        self.env = T.BatchedContainer(B, cs, n, reward, "diff", packing_strategy=strategy, device=device)
        self.dyn = [torch.empty_like(self.dynamic0[0]), torch.empty_like(self.dynamic0[0])]
        self.csb = [torch.empty_like(self.cs0[0]), torch.empty_like(self.cs0[0])]
        self.bitb = [torch.empty_like(self.bits0[0]), torch.empty_like(self.bits0[0])] if self.bits else None
        self.mask0 = torch.ones(B, self.nR, **f32)
        self.maskb = [torch.empty(B, self.nR, **f32), torch.empty(B, self.nR, **f32)]
        self.cur = torch.empty(B, self.nR, **f32)
        self.feat = torch.empty(self.env._feature_shape(), **f32)
        self.reward = torch.empty(B, **f32)
        self.lib = _lib.lib()
        self.ctx = _lib.ctx(device)
        self.hook = None                                     # optional per-kernel timing hook
        self.fused = fused

    def _k(self, name, fn, *args):
        stream = _lib.stream_of(self.device)
        if self.hook:
            self.hook(name, lambda: _lib.check(fn(*args, stream), self.ctx))
        else:
            _lib.check(fn(*args, stream), self.ctx)

    def episode(self):
        """fused: n launches of tap_transition (first FRESH, last emits calc_ratio);
        otherwise reset + n x (tap_mask_step, tap_env_step_gather) + tap_env_ratio."""
        L, P, e = self.lib, _lib.ptr, self.env
        d = C.byref(e.desc)
        if not self.fused:
            self._k("reset", L.tap_env_reset, self.ctx, d, P(e._state))
        step = 0
        for w in range(self.windows):
            st = self.static[w]
            dyn_in, cs_in, mask_in = self.dynamic0[w], self.cs0[w], self.mask0
            bits_in = self.bits0[w]
            for t in range(self.nw):
                ptr = self.tape[w][t]
                o = t & 1
                flags = (_lib.TAP_T_FRESH if step == 0 else 0) | (_lib.TAP_T_RATIO if step == self.n - 1 else 0)
                if self.fused and self.bits:
                    self._k("transition", L.tap_transition_bits, self.ctx, d, P(e._state), self.nw, self.R,
                            self.rows, 3, P(bits_in), P(st), st.shape[1], P(ptr), P(mask_in), P(self.bitb[o]),
                            P(self.dyn[o]), P(self.cur), P(self.maskb[o]), P(self.feat), P(self.reward), flags)
                    bits_in = self.bitb[o]
                elif self.fused:
                    self._k("transition", L.tap_transition, self.ctx, d, P(e._state), self.nw, self.R, self.rows,
                            3, P(dyn_in), P(st), st.shape[1], P(ptr), P(mask_in), P(cs_in), P(self.dyn[o]),
                            P(self.csb[o]), P(self.cur), P(self.maskb[o]), P(self.feat), P(self.reward), flags)
                elif self.bits:
                    self._k("mask_step", L.tap_mask_step_bits, self.ctx, self.B, self.nw, self.R, self.rows, 3,
                            P(bits_in), P(st), st.shape[1], P(ptr), P(mask_in), P(self.bitb[o]), P(self.dyn[o]),
                            P(self.cur), P(self.maskb[o]))
                    self._k("env_step", L.tap_env_step_gather, self.ctx, d, P(e._state), P(st),
                            st.shape[1], self.nR, P(ptr), None, P(self.feat))
                    bits_in = self.bitb[o]
                else:
                    self._k("mask_step", L.tap_mask_step, self.ctx, self.B, self.nw, self.R, self.rows, 3,
                            P(dyn_in), P(st), st.shape[1], P(ptr), P(mask_in), P(cs_in),
                            P(self.dyn[o]), P(self.csb[o]), P(self.cur), P(self.maskb[o]))
                    self._k("env_step", L.tap_env_step_gather, self.ctx, d, P(e._state), P(st),
                            st.shape[1], self.nR, P(ptr), None, P(self.feat))
                dyn_in, cs_in, mask_in = self.dyn[o], self.csb[o], self.maskb[o]
                step += 1
        if not self.fused:
            self._k("ratio", L.tap_env_ratio, self.ctx, d, P(e._state), P(self.reward), None, None)


class RollingHotPath(HotPath):
    """c5: true rolling windows over device-generated 50-block instances (initial container
    7x7x250 as scripts/rolling.sh); actions replayed from a tape recorded with a random feasible
    policy."""

    def __init__(self, cfg, B, start, device, seed=12345, window=10, fused_rolling=False, overlap=False):
        _, D, cs, n, _, reward, strategy = cfg
        self.overlap = overlap
        self.side = torch.cuda.Stream(device=device)
        self.D, self.cs, self.n, self.B, self.device, self.nw = D, cs, n, B, device, window
        self.fused, self.hook, self.rolling, self.fused_rolling = True, None, True, fused_rolling
        self.lib, self.ctx = _lib.lib(), _lib.ctx(device)
        init = [7, 250] if D == 2 else [7, 7, 250]
        _, _, blocks, positions = T.generate.generate_instances(B, n, D, init[0], init[-1], 1, (1, 5),
                                                                seed=seed + start, device=device, return_aux=True)
        g = torch.Generator(device=device)
        g.manual_seed(seed + 1 + start)
        rec = T.run_rolling_episode(blocks, positions, init,
                                    lambda current_mask, **_: torch.multinomial(current_mask, 1, generator=g).squeeze(1),
                                    cs[0], cs[-1], child_graph_size=window, reward_type=reward)
        rec["env"].check()
        self.tape = rec["tour_idx"].t().contiguous()             # (n, B)
        self.rw = rec["windows"]                                 # holds blocks + relation masks
        self.env = T.BatchedContainer(B, cs, n, reward, "diff", packing_strategy=strategy, device=device)
        self.R = 2 if D == 2 else 6
        self.nR, self.rows = window * self.R, 3 * window
        f32 = dict(dtype=torch.float32, device=device)
        self.static = torch.empty(B, 1 + D, self.nR, **f32)
        self.static2 = torch.empty(B, 1 + D, self.nR, **f32)
        self.dyn = [torch.empty(B, self.rows, self.nR, **f32) for _ in range(3)]
        self.csb = [torch.empty(B, 3, self.nR, **f32) for _ in range(3)]
        self.mask0 = torch.ones(B, self.nR, **f32)
        self.maskb = [torch.empty(B, self.nR, **f32), torch.empty(B, self.nR, **f32)]
        self.cur = torch.empty(B, self.nR, **f32)
        self.feat = torch.empty(self.env._feature_shape(), **f32)
        self.reward = torch.empty(B, **f32)
        self.state = torch.zeros(B, 2, dtype=torch.int64, device=device)
        self.bits = T.pack.bits_supported(self.rows, self.nR)    # the windows' bit shadow (tap_rolling_window emits it)
        self.bitb = [torch.empty(B, self.nR, dtype=torch.int64, device=device) for _ in range(3)] if self.bits else [None] * 3
        self.want = rec["reward"].clone()

    def episode(self):
        L, P, e, rw = self.lib, _lib.ptr, self.env, self.rw
        d = C.byref(e.desc)
        self.state.zero_()
        self._k("reset", L.tap_env_reset, self.ctx, d, P(e._state))
        n1 = self.n - self.nw
        st = [self.static, self.static2]
        cs2 = None if self.bits else self.csb[2]          # the column sums are only needed without the bit shadow
        self._k("rolling_window", L.tap_rolling_window, self.ctx, self.B, self.D, self.n, self.nw,
                P(rw.blocks), P(rw.rel), P(self.state), None, P(st[0]), P(self.dyn[2]), P(cs2),
                P(self.bitb[2]), P(self.cur), None, None)
        prev_ev = None
        for t in range(n1):
            if self.fused_rolling:                                # placement t + window t+1: one launch
                self._k("rolling_step", L.tap_rolling_step, self.ctx, d, P(e._state), self.n, self.nw, P(rw.blocks),
                        P(rw.rel), P(self.state), P(self.tape[t]), P(st[t & 1]), P(st[(t + 1) & 1]), P(self.dyn[2]),
                        P(cs2), P(self.bitb[2]), P(self.cur), None, None, P(self.feat))
            elif not self.overlap:
                self._k("env_step", L.tap_env_step_gather, self.ctx, d, P(e._state), P(st[t & 1]),
                        self.static.shape[1], self.nR, P(self.tape[t]), None, P(self.feat))
                self._k("rolling_window", L.tap_rolling_window, self.ctx, self.B, self.D, self.n, self.nw,
                        P(rw.blocks), P(rw.rel), P(self.state), P(self.tape[t]), P(st[(t + 1) & 1]), P(self.dyn[2]),
                        P(cs2), P(self.bitb[2]), P(self.cur), None, None)
            else:
                # placement t (reads window t's static, owns the container state) and window t+1 (owns
                # the window state, writes the OTHER static buffer) both depend only on the pick of step
                # t: two HIP streams, so the latency-bound placement runs under the write-bound window
                # kernel.  window t+1 overwrites the static buffer placement t-1 read: wait for that one.
                main = torch.cuda.current_stream(self.device)
                self.side.wait_stream(main)
                with torch.cuda.stream(self.side):
                    self._k("env_step", L.tap_env_step_gather, self.ctx, d, P(e._state), P(st[t & 1]),
                            self.static.shape[1], self.nR, P(self.tape[t]), None, P(self.feat))
                    ev = torch.cuda.Event()
                    ev.record(self.side)
                if prev_ev is not None:
                    main.wait_event(prev_ev)
                prev_ev = ev
                self._k("rolling_window", L.tap_rolling_window, self.ctx, self.B, self.D, self.n, self.nw,
                        P(rw.blocks), P(rw.rel), P(self.state), P(self.tape[t]), P(st[(t + 1) & 1]), P(self.dyn[2]),
                        P(cs2), P(self.bitb[2]), P(self.cur), None, None)
        if self.overlap and not self.fused_rolling:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
        self.static_last = st[n1 & 1]
        dyn_in, cs_in, mask_in, bits_in = self.dyn[2], self.csb[2], self.mask0, self.bitb[2]
        for t in range(self.nw):                                  # the last graph: a whole episode
            o = t & 1
            flags = _lib.TAP_T_RATIO if t == self.nw - 1 else 0
            if self.bits:
                self._k("transition", L.tap_transition_bits, self.ctx, d, P(e._state), self.nw, self.R, self.rows, 3,
                        P(bits_in), P(self.static_last), self.static.shape[1], P(self.tape[n1 + t]), P(mask_in),
                        P(self.bitb[o]), P(self.dyn[o]), P(self.cur), P(self.maskb[o]), P(self.feat), P(self.reward), flags)
            else:
                self._k("transition", L.tap_transition, self.ctx, d, P(e._state), self.nw, self.R, self.rows, 3,
                        P(dyn_in), P(self.static_last), self.static.shape[1], P(self.tape[n1 + t]), P(mask_in), P(cs_in),
                        P(self.dyn[o]), P(self.csb[o]), P(self.cur), P(self.maskb[o]), P(self.feat), P(self.reward), flags)
            dyn_in, cs_in, mask_in, bits_in = self.dyn[o], self.csb[o], self.maskb[o], self.bitb[o]


GATHER_EVERY = 8   # passes whose (B,) reward vectors share one RCCL all-gather (fewer, larger collectives)


def time_passes(hp, steps, warmup, use_graph, world):
    dev = hp.device
    handles = []
    import torch.distributed as tdd
    # TAP_BENCH_FORCE_GATHER=1 runs the N > 1 code path (accumulate + async all-gather) in a 1-rank
    # process group: the only way to time its overhead on a 1-GPU box
    gather = world > 1 or (os.environ.get("TAP_BENCH_FORCE_GATHER") == "1" and tdd.is_available() and tdd.is_initialized())
    nranks = tdd.get_world_size() if gather else 1
    acc = torch.empty(GATHER_EVERY, hp.B, dtype=torch.float32, device=dev) if gather else None
    state = {"i": 0}

    def flush(count):
        import torch.distributed as dist
        buf = acc[:count].clone()                              # the only exchange of the whole job
        out = [torch.empty_like(buf) for _ in range(nranks)]
        handles.append((dist.all_gather(out, buf, async_op=True), out))

    def one_pass(g):
        if g is not None:
            g.replay()
        else:
            hp.episode()
        if gather:
            acc[state["i"]].copy_(hp.reward)
            state["i"] += 1
            if state["i"] == GATHER_EVERY:
                flush(GATHER_EVERY)
                state["i"] = 0

    def drain():
        if gather and state["i"]:
            flush(state["i"])
            state["i"] = 0
        for h, _ in handles:
            h.wait()
        handles.clear()

    graph = None
    if use_graph:
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            hp.episode()                                        # warm the allocator / lazy init
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            hp.episode()
    for _ in range(warmup):
        one_pass(graph)
    drain()
    tdist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        one_pass(graph)
    drain()
    torch.cuda.synchronize(dev)
    tdist.barrier()
    dt = time.perf_counter() - t0
    return tdist.max_over_ranks(dt, dev), graph


def kernel_event_times(hp, steps, graph=None):
    """Per-launch durations with HIP events on the launch stream (torch.cuda.Event on torch's
    current stream == the stream the kernels are enqueued on), over `steps` eager passes."""
    recs = {}

    def hook(name, launch):
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        launch()
        b.record()
        recs.setdefault(name, []).append((a, b))

    pass_pairs = []
    pure = hp.fused and not getattr(hp, 'rolling', False)
    hp.hook = None if pure else hook
    try:
        for _ in range(steps):
            if pure:
                # the pass is n back-to-back launches of ONE kernel: bracket the pass and divide, so the
                # event overhead is amortised and the figure is comparable with rocprofv3's average
                a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
                a.record()
                if graph is not None:
                    graph.replay()
                else:
                    hp.episode()
                b.record()
                pass_pairs.append((a, b))
            else:
                hp.episode()
        torch.cuda.synchronize(hp.device)
    finally:
        hp.hook = None
    if pure:
        us = np.array([a.elapsed_time(b) for a, b in pass_pairs]) * 1e3 / hp.n
        recs = None
    # cost of an empty event pair, to show how much of a short kernel's figure is event overhead
    pairs = []
    for _ in range(64):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); b.record(); pairs.append((a, b))
    torch.cuda.synchronize(hp.device)
    empty_us = float(np.median([a.elapsed_time(b) for a, b in pairs]) * 1e3)
    out = {}
    if recs is None:
        out["transition"] = dict(launches=len(us) * hp.n, avg_us=float(us.mean()), med_us=float(np.median(us)),
                                 total_us=float(us.sum() * hp.n))
        return out, empty_us
    for name, evs in recs.items():
        us = np.array([a.elapsed_time(b) for a, b in evs]) * 1e3
        out[name] = dict(launches=len(us), avg_us=float(us.mean()), med_us=float(np.median(us)),
                         total_us=float(us.sum()))
    return out, empty_us


def cpu_baseline(cfg, window=None, budget_s=12.0, instances=None):
    """The oracle (C port of the reference algorithm) over the same pass, on the host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    _, D, cs, n, B, reward, strategy = cfg
    B = min(B, 4096)
    nw = window or n
    wins = []
    for w in range(n // nw):
        if isinstance(instances, tuple) and len(instances) == 3:     # ("given", [static per window], [dynamic per window])
            static, dynamic = instances[1][w][:B].cpu(), instances[2][w][:B].cpu()
        elif instances is not None:
            static, dynamic = synth.tiled_instances(instances[0], instances[1], B)
        else:
            static, dynamic = synth.rand_instances(B, nw, D, seed=12345 + 100 * w)
        tape = synth.random_feasible_tape(static, dynamic, nw, seed=12346 + 100 * w).numpy()
        wins.append((static.numpy(), dynamic.numpy(), tape))
    R = wins[0][0].shape[2] // nw
    blocks = np.concatenate([np.stack([st[np.arange(B), 1:, tape[:, t]] for t in range(nw)], axis=1)
                             for st, _, tape in wins], axis=1).astype(np.int32)
    desc = O.make_desc(cs, n, reward, "diff", strategy)
    O.lib()
    done, t0 = 0, time.perf_counter()
    while True:
        for st, dyn0, tape in wins:
            dyn, mask = dyn0, np.ones((B, st.shape[2]), np.float32)
            O.initial_mask(dyn, nw)
            for t in range(nw):
                dyn = O.update_dynamic(dyn, st, tape[:, t], nw, 3)
                _, mask = O.update_mask(mask, dyn, tape[:, t], nw, R)
        r = O.run_episodes(desc, blocks, nthreads=1, want_heightmaps=False)
        assert r["nerr"] == 0
        done += B * n
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    return dict(value=done / el, unit="env-steps/s", cores=1, kind="port",
                sample="%d passes of B=%d envs x n=%d (masks + placements + ratio), %.1f s, "
                       "oracle/libtap_oracle.so single thread" % (done // (B * n), B, n, el))


def cpu_baseline_rolling(cfg, window, budget_s=12.0):
    """The oracle over rolling.validate's loop (windows re-cut after every placement), one host core."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    _, D, cs, n, B, reward, strategy = cfg
    init = [7, 250] if D == 2 else [7, 7, 250]
    rng = np.random.RandomState(1)
    blocks = synth.rand_blocks(64, n, D, seed=777)
    inst = []
    for b in range(64):                      # instances = blocks the oracle's generator restatement accepts
        rc, pos, _, _ = O.instance_from_blocks(blocks[b], init, 1)
        if rc == 1:
            inst.append((blocks[b], pos))
    done, t0 = 0, time.perf_counter()
    while True:
        for bl, pos in inst:
            ro = O.Rolling(bl, pos, init, window)
            e = O.Env(cs, n, reward, "diff", strategy)
            for t in range(n - window + 1):
                rc, st, dy, nodes = ro.convert_to_input()
                cur = O.initial_mask(dy[None], window)[0]
                if t == n - window:
                    break
                p = int(rng.choice(np.flatnonzero(cur)))
                e.add_new_block(st[1:, p])
                ro.remove(p % window)
            mask = np.ones((1, st.shape[1]), np.float32)
            dyn = dy[None]
            for t in range(window):
                p = np.array([rng.choice(np.flatnonzero(cur))], dtype=np.int64)
                e.add_new_block(st[1:, int(p[0])])
                dyn = O.update_dynamic(dyn, st[None], p, window, 3)
                c2, mask = O.update_mask(mask, dyn, p, window, st.shape[1] // window)
                cur = c2[0]
            e.calc_ratio()
            done += n
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    return dict(value=done / el, unit="env-steps/s", cores=1, kind="port",
                sample="%d rolling episodes of %d placements (window %d) over %d oracle-generated instances, %.1f s, "
                       "oracle/libtap_oracle.so single thread driven per step from Python" % (done // n, n, window, len(inst), el))


def load_traffic(name):
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get(name)
        except Exception:
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="two launches per step (mask_step, env_step) + reset + ratio")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--synthetic-precedence", action="store_true",
                    help="c2/c3: RAND-marginal blocks with a random precedence DAG (synth.rand_instances) instead of "
                         "instances from the device-side RAND generator (blocks packed into the 7-wide initial container)")
    ap.add_argument("--rand-blocks", action="store_true",
                    help="c4: RAND-marginal synthetic instances instead of the reference-generated PPSG fixture tiled x128")
    ap.add_argument("--no-bits", action="store_true",
                    help="precedence update as an fp32 copy (tap_transition) instead of on the bit shadow (tap_transition_bits)")
    ap.add_argument("--sweep", action="store_true", help="also print a batch sweep to stderr")
    ap.add_argument("--two-launch-rolling", action="store_true",
                    help="c5: tap_env_step_gather + tap_rolling_window per step instead of the fused tap_rolling_step")
    ap.add_argument("--overlap", action="store_true",
                    help="c5: placement t and window t+1 on two HIP streams (measured slower on this stack: "
                         "285 vs 297 M env-steps/s in a graph, 251 vs 299 eager -- the cross-stream waits cost more "
                         "than the 8.6 us placement they hide)")
    ap.add_argument("--approx-windows", action="store_true",
                    help="c5: consecutive independent 10-node windows instead of true rolling windows")
    args = ap.parse_args()

    rank, world, local = tdist.init_from_env()
    if world != args.gpus and rank == 0 and world > 1:
        print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: libtapenv has no CPU path")
    local = local % torch.cuda.device_count()   # > 1 rank per GPU only happens in the gloo self-test
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cfg = CONFIGS[args.config]
    name, D, cs, n, B, reward, strategy = cfg
    if args.batch:
        B = args.batch
        cfg = (name, D, cs, n, B, reward, strategy)
    rolling = args.config in ROLLING and not args.approx_windows
    if rolling:
        hp = RollingHotPath(cfg, B, rank * B, dev, window=WINDOW[args.config], fused_rolling=not args.two_launch_rolling,
                            overlap=args.overlap)
    else:
        instances = None
        if args.config in ("c2", "c3") and not args.synthetic_precedence:
            instances = "generate"
        if args.config == "c4" and not args.rand_blocks:
            fx = os.path.join(ROOT, "tests", "golden", "ppsg_2d.npz")
            if os.path.exists(fx):                                  # 64 PPSG instances written by the reference
                z = np.load(fx)                                     # (tests/golden/make_golden.py --only ppsg)
                instances = (z["static"].astype(np.float32), z["dynamic"].astype(np.float32))
                name = name.replace("RAND-marginal blocks", "the reference's PPSG generator: 64 instances tiled x%d" % (B // 64))
                cfg = (name, D, cs, n, B, reward, strategy)
        hp = HotPath(cfg, B, rank * B, dev, fused=not args.unfused, window=WINDOW.get(args.config), bits=not args.no_bits,
                     instances=instances)
    use_graph = not args.no_graph
    dt, graph = time_passes(hp, args.steps, args.warmup, use_graph, world)
    hp.env.check()
    total_steps = B * world * n * args.steps
    value = total_steps / dt

    out = None
    if rank == 0:
        kt, empty_us = kernel_event_times(hp, max(3, min(args.steps, 20)), graph if (hp.fused and not rolling) else None)
        env_b, mask_b = algorithmic_bytes(D, cs, hp.nw)
        R_ = 2 if D == 2 else 6
        win_b = (1 + D) * hp.nw * R_ * 4 + 3 * hp.nw * hp.nw * R_ * 4 + hp.nw * R_ * 4 + 32   # static + dynamic + mask + state
        per_launch = {"env_step": env_b * B, "mask_step": mask_b * B, "transition": (env_b + mask_b) * B,
                      "rolling_window": win_b * B, "rolling_step": (win_b + env_b) * B}
        names = [k for k in ("transition", "rolling_step", "rolling_window", "mask_step", "env_step", "ratio", "reset") if k in kt]
        dom = max([k for k in names if k in per_launch], key=lambda k: kt[k]["total_us"])
        ach = per_launch[dom] / (kt[dom]["avg_us"] * 1e-6) / 1e9
        npass = max(3, min(args.steps, 20))
        kernels = {}
        for k in names:
            kernels[k] = dict(avg_us=round(kt[k]["avg_us"], 3), launches_per_pass=kt[k]["launches"] // npass)
            if k in per_launch:
                kernels[k]["alg_bytes_per_launch"] = per_launch[k]
                kernels[k]["alg_GBps"] = round(per_launch[k] / (kt[k]["avg_us"] * 1e-6) / 1e9, 2)
        out = {
            "metric": "env-steps/s (batch placements) 2D n=10 LB_GREEDY; 1/2/4/8 GPU + CPU ref",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "i32 height-maps / f64 candidate scores / f32 masks",
            "data": "synthetic",
            "config": {"workload": name, "batch_per_gpu": B, "nodes": n, "window_nodes": hp.nw, "container": cs,
                       "reward_type": reward, "packing_strategy": strategy,
                       "instances": ("device-generated 50-block RAND instances (generate.generate_instances), initial container 7 wide"
                                     if rolling else "RAND instances of the device-side generator (generate_blocks semantics: "
                                     "random blocks packed into the 7-wide initial container, real precedence)"
                                     if getattr(hp, "instances", None) == "generate" else
                                     "fixture tiled" if getattr(hp, "instances", None) is not None else
                                     "RAND-marginal blocks, random precedence DAG (synth.rand_instances)"),
                       "pass": (("rolling.validate's loop: (n - window) x tap_rolling_step (placement t + window t+1 in one launch), "
                                 "then window x tap_transition_bits on the last graph") if getattr(hp, "fused_rolling", False) else
                                ("rolling.validate's loop: (n - window) x (tap_env_step_gather + tap_rolling_window), "
                                 "then window x tap_transition_bits on the last graph")) if rolling else
                               ("n x tap_transition%s (update_dynamic+update_mask+gather+add_new_block in one launch; "
                                "first starts a fresh container, last emits calc_ratio)%s" %
                                (("_bits", "; dynamic carried between steps as a bit shadow, the fp32 tensor is written "
                                  "every step but not re-read") if getattr(hp, "bits", False) else ("", ""))) if hp.fused else
                               "reset + n x (update_dynamic+update_mask, add_new_block) + calc_ratio",
                       "launch": "hipGraph replay" if use_graph else "eager"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": load_traffic(args.config + ":" + dom + ("_copy" if dom == "transition" and not rolling and not getattr(hp, "bits", False) else "")),
                         "alg_bytes_per_env_step": per_launch[dom] // B,
                         "units_per_launch": B, "avg_launch_us": kt[dom]["avg_us"],
                         "event_pair_overhead_us": empty_us},
            "kernels": kernels,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline_rolling(cfg, WINDOW[args.config]) if rolling else cpu_baseline(cfg, WINDOW.get(args.config), instances=("given", hp.static, hp.dynamic0) if getattr(hp, 'instances', None) == 'generate' else getattr(hp, 'instances', None))
        if args.sweep:
            for b in (8192, 32768, 131072, 524288, 2097152):
                try:
                    h2 = HotPath((name, D, cs, n, b, reward, strategy), b, 0, dev, fused=hp.fused, window=hp.nw)
                    d2, _ = time_passes(h2, 10, 3, use_graph, 1)
                    k2, _ = kernel_event_times(h2, 3)
                    print("sweep B=%d: %.3e env-steps/s; " % (b, b * n * 10 / d2) + "; ".join(
                        "%s %.1f us (%.0f GB/s alg)" % (k, v["avg_us"], per_launch[k] / B * b / v["avg_us"] / 1e3)
                        for k, v in k2.items() if k in per_launch), file=sys.stderr)
                    del h2
                except Exception as ex:  # out of memory at the top end is fine
                    print("sweep B=%d failed: %s" % (b, ex), file=sys.stderr)
                    break
        cb = out.get("cpu_baseline")
        rp = os.path.join(ROOT, "profiles", "r01d_reference_cpu.jsonl")
        if cb and os.path.exists(rp):
            # the reference's own Python path cannot run on the GPU box; scripts/time_reference.py measured, in
            # the build container, how much slower it is than the oracle on the same core and the same work
            key = {"c2": "c1/c2", "c3": "c3", "c4": "c4"}.get(args.config)
            for line in open(rp):
                r = json.loads(line)
                if r.get("config") == key:
                    cb["reference_derived"] = dict(
                        value=cb["value"] / r["ratio_oracle_over_reference"], unit="env-steps/s", cores=1,
                        how="this oracle figure / %.0f (reference vs oracle on one core of the build container, "
                            "profiles/r01d_reference_cpu.jsonl); derived, not measured here" % r["ratio_oracle_over_reference"])
        print(json.dumps(out))
    tdist.barrier()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
