#!/usr/bin/env python3
"""Benchmark of the batched Transport-and-Pack environment hot path on MI355X.

One "step" = one full pass of the hot path over one batch of synthetic instances, i.e. what
DRL.forward does around its policy network for one batch (model.py:294-515), with the actions
replayed from a pre-computed feasible tape:

    tap_transition_first (step 0: reads the instance batch's fp32 `dynamic` and builds its bit shadow INSIDE
       the launch, starts a fresh container)
    -> (n - 1) x tap_transition_bits (update_dynamic + update_mask + gather + add_new_block in one launch;
       the last emits calc_ratio)
    [-> when N > 1, the (B,) reward vectors of 8 consecutive passes are all-gathered with one RCCL call]

value = env-steps/s = (placements in all envs on all ranks) / wall time, inputs resident in HBM before
the timed region.  After the timed region (outside it) the outputs of the LAST replayed pass are
compared with the CPU oracle on a slice of the batch ("verified"); a mismatch exits non-zero.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c3|c4|c5|c6|k6] [--batch B]
                    [--no-graph] [--no-cpu-baseline] [--no-variants] [--sweep [--sweep-out FILE]]

--gpus N > 1: launched by torch.distributed.run (one rank per GPU, RCCL) the ranks find each other
through RANK / WORLD_SIZE / MASTER_*; launched as plain `python bench.py --gpus N` the script spawns the N
ranks itself (torch.multiprocessing).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import socket
import sys
import statistics
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import tap_net_amd as T                     # noqa: E402
from tap_net_amd import _lib, synth         # noqa: E402
from tap_net_amd import dist as tdist       # noqa: E402



def _usable_cpus():
    """CPUs this process may use (affinity and cgroup quota), capped: the oracle's OpenMP team for verification"""
    return max(1, min(cpu_info()[2], 32))


def _oracle():
    """The CPU oracle (test infrastructure): only the verification after the timed region and the
    cpu_baseline leg load it -- never the timed path."""
    tests = os.path.join(ROOT, "tests")
    if tests not in sys.path:
        sys.path.insert(0, tests)
    import oracle_lib
    return oracle_lib


HBM_PEAK_GBS = 8000.0    # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md; measured ceilings: load_calibration()
VERIFY_ENVS = 512        # envs of the last replayed pass compared with the oracle after the timed region

# nodes of one precedence window (rolling.py feeds the actor 10-node sub-graphs)
WINDOW = {"c5": 10}
# configs whose pass is rolling.validate's loop: N - 10 one-step windows re-cut from the precedence DAG
# after every placement, then a full episode on the last one
ROLLING = {"c5"}

CONFIGS = {
    # name: (workload string, D, container, n, per-GPU batch, reward, strategy)
    "c2": ("2D RAND nodes=10 container_width=5 LB_GREEDY batch=8192 on 1xMI355X (BASELINE configs[1])",
           2, [5, 50], 10, 8192, "C+P+S-lb-soft", "LB_GREEDY"),
    "c3": ("3D RAND nodes=10 container_width=5x5 LB_GREEDY batch=4096 on 1xMI355X (BASELINE configs[2])",
           3, [5, 5, 50], 10, 4096, "C+P+S-lb-soft", "LB_GREEDY"),
    "c4": ("2D nodes=20 container_width=7 MACS batch=8192 on 1xMI355X (BASELINE configs[3], RAND-marginal blocks)",
           2, [7, 100], 20, 8192, "C+P+S-mcs-soft", "MACS"),
    "c5": ("3D MIX nodes=50 (rolling 10-node windows) container_width=5x5 H=250 LB_GREEDY "
           "batch=8192 per GPU (BASELINE configs[4] shard)",
           3, [5, 5, 250], 50, 8192, "C+P+S-lb-soft", "LB_GREEDY"),
    "c6": ("3D RAND nodes=10 container_width=5x5 MACS batch=4096 on 1xMI355X (not a BASELINE config: SURVEY 8(f) f3)",
           3, [5, 5, 50], 10, 4096, "C+P+S-mcs-soft", "MACS"),
    "c7": ("3D RAND nodes=10 container_width=10x10 LB_GREEDY batch=4096 on 1xMI355X (not a BASELINE config: the "
           "wave-per-container kernels, model.py:279 builds W x W for any --container_width)",
           3, [10, 10, 50], 10, 4096, "C+P+S-lb-soft", "LB_GREEDY"),
    "c8": ("3D RAND nodes=10 container_width=10x10 MACS batch=4096 on 1xMI355X (not a BASELINE config: the "
           "wave-per-container MACS 3D kernel, two launches per step)",
           3, [10, 10, 50], 10, 4096, "C+P+S-mcs-soft", "MACS"),
    "c9": ("2D RAND nodes=10 container_width=100 MACS batch=4096 on 1xMI355X (not a BASELINE config: the "
           "wave-per-container MACS 2D kernel, two launches per step)",
           2, [100, 50], 10, 4096, "C+P+S-mcs-soft", "MACS"),
    "k6": ("2D RAND nodes=10 container_width=5 LB_GREEDY batch=8192: whole episode per launch "
           "(pack.reward / calc_positions_lb_greedy, SURVEY K6; not a BASELINE config)",
           2, [5, 50], 10, 8192, "C+P+S-lb-soft", "LB_GREEDY"),
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


def compulsory_bytes(kind, D, cs, n, bits=True, static_rows=None):
    """Bytes ONE launch of the implemented kernel must move per env, whatever the cache does:
    what it reads that it has not produced itself plus what it writes.  This is what roofline.achieved prices;
    SURVEY's per-env-step figure (which prices an fp32 read of `dynamic` the bit-shadow step never performs) is
    kept beside it as frac_alg_survey."""
    cells = int(np.prod(cs[:-1]))
    R = 2 if D == 2 else 6
    nR, rows = n * R, 3 * n
    flen = cs[0] - 1 if D == 2 else 2 * cells
    # container: hm + 4 counters in and out, position + stable flag out, feature out, the block's D sides in
    env = 2 * (cells * 4 + 16) + D * 4 + 1 + flen * 4 + D * 4
    # precedence update on the bit shadow: ptr, ONE float of row 0 of static (the block id of the picked column; the
    # stream wave fetches the whole 4 nR-byte row and shuffles, but 4 bytes are what the step needs), mask in; shadow
    # in + out; fp32 tensor out; both masks out
    shadow = 8 + 4 + nR * 4 + 2 * nR * 8 + rows * nR * 4 + 2 * nR * 4
    copy = 8 + 4 + nR * 4 + 2 * 3 * nR * 4 + 2 * rows * nR * 4 + 2 * nR * 4     # column-sum shadow in + out, tensor in + out
    if kind == "mask_step":
        return shadow if bits else copy
    if kind == "env_step":
        return env
    if kind == "transition":
        return env + (shadow if bits else copy)
    if kind == "transition_first":                  # reads the fresh fp32 tensor once, no shadow in
        return env + shadow - nR * 8 + rows * nR * 4
    if kind == "dyn_bits":
        return rows * nR * 4 + nR * 8
    if kind in ("rolling_window", "rolling_step"):
        # in: the window nodes' five relation masks and block sides, the 2-word window state (in + out), ptr;
        # out: static (1+D, nR), dynamic (3 child, nR), its shadow, the initial mask, the node ids
        win = n * 5 * 8 + n * D * 4 + 2 * 16 + 8 + (1 + D) * nR * 4 + rows * nR * 4 + nR * 8 + nR * 4 + n * 4
        return win + (env if kind == "rolling_step" else 0)
    raise KeyError(kind)


def load_calibration():
    """Measured ceilings of this part (scripts/calibrate_bw.py, the hot kernels' access shape), newest profile set."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bw_calibration.json")))
    if not files:
        return None
    with open(files[-1]) as f:
        c = json.load(f)
    c.pop("rows", None)
    c["source"] = "profiles/" + os.path.basename(files[-1])
    return c


def episode_bytes(D, n):
    """SURVEY.md 8(d), fused full episode (K6): blocks n*D*4 (+ tour n*8) read, pos n*D*4 + stable n + reward 4
    written -- 254 B per 2D n=10 episode."""
    return n * D * 4 + n * 8 + n * D * 4 + n + 4


class HotPath(object):
    """Pre-allocated buffers + direct C-ABI calls for one rank's share of the batch.

    An episode is `windows` consecutive precedence windows of `nw` nodes each over ONE long-lived
    container per env (windows = 1 except for --approx-windows)."""

    rolling = False
    kind = "transition"

    def __init__(self, cfg, B, start, device, seed=12345, fused=True, window=None, bits=True, instances=None, tile_from=None):
        _, D, cs, n, _, reward, strategy = cfg
        self.cfg = cfg
        self.D, self.cs, self.n, self.B, self.device = D, cs, n, B, device
        self.reward_type, self.strategy = reward, strategy
        self.nw = window or n
        assert n % self.nw == 0
        self.windows = n // self.nw
        f32 = dict(dtype=torch.float32, device=device)
        self.instances = instances
        self.static, self.dynamic0, self.tape, self.cs0 = [], [], [], []
        for w in range(self.windows):
            if tile_from is not None:                               # batch sweep: another pass's instances and tape, tiled
                k = B // tile_from.B
                assert k * tile_from.B == B
                self.static.append(tile_from.static[w].repeat(k, 1, 1))
                self.dynamic0.append(tile_from.dynamic0[w].repeat(k, 1, 1))
                self.tape.append(tile_from.tape[w].repeat(1, k).contiguous())
                self.cs0.append(T.pack.dynamic_colsum(self.dynamic0[-1], self.nw).clone() if not bits else None)
                continue
            if instances == "ppsg2d":                               # PPSG instances of the device-side generator
                t0 = time.perf_counter()
                static, dynamic = synth.device_ppsg_instances(B, self.nw, D, seed=seed + 100 * w, start=start, device=device,
                                                              target_container_width=cs[0])
                torch.cuda.synchronize(device)
                self.generate_s = time.perf_counter() - t0
            elif instances == "generate":                           # RAND instances of the device-side generator (f1)
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
            self.cs0.append(T.pack.dynamic_colsum(self.dynamic0[-1], self.nw).clone() if not bits else None)
        self.R = self.static[0].shape[2] // self.nw
        self.nR, self.rows = self.static[0].shape[2], self.dynamic0[0].shape[1]
        # the bit shadow is only valid for 0/1 tensors of a supported shape: checked once here (setup), the
        # shadow itself is rebuilt from the fp32 tensor inside every pass (episode())
        self.bits = bool(bits and T.pack.bits_supported(self.rows, self.nR))
        if self.bits:
            self.bits = all(int(T.pack.dynamic_bits(d)[1].item()) == 0 for d in self.dynamic0)
        if not self.bits and self.cs0[0] is None:
            self.cs0 = [T.pack.dynamic_colsum(d, self.nw).clone() for d in self.dynamic0]
        self.env = T.BatchedContainer(B, cs, n, reward, "diff", packing_strategy=strategy, device=device)
        self.dyn = [torch.empty_like(self.dynamic0[0]), torch.empty_like(self.dynamic0[0])]
        self.csb = [torch.empty(B, 3, self.nR, **f32), torch.empty(B, 3, self.nR, **f32)] if not self.bits else [None, None]
        i64 = dict(dtype=torch.int64, device=device)
        self.bitb = [torch.empty(B, self.nR, **i64) for _ in range(3)] if self.bits else None
        self.nonbin = torch.zeros(1, dtype=torch.int32, device=device)
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

    def launches_per_pass(self):
        per = 1 if self.fused else 2
        return self.n * per + (self.windows if self.bits and not self.fused else 0) + (0 if self.fused else 2)

    def episode(self):
        """fused: [tap_dyn_bits +] n launches of tap_transition (first FRESH, last emits calc_ratio);
        otherwise reset + n x (tap_mask_step, tap_env_step_gather) + tap_env_ratio."""
        L, P, e = self.lib, _lib.ptr, self.env
        d = C.byref(e.desc)
        if not self.fused:
            self._k("reset", L.tap_env_reset, self.ctx, d, P(e._state))
        step = 0
        for w in range(self.windows):
            st = self.static[w]
            dyn_in, cs_in, mask_in = self.dynamic0[w], self.cs0[w], self.mask0
            bits_in = None
            if self.bits and not self.fused:
                bits_in = self.bitb[2]
                self._k("dyn_bits", L.tap_dyn_bits, self.ctx, self.B, self.nR, self.rows, P(dyn_in), P(bits_in), P(self.nonbin))
            for t in range(self.nw):
                ptr = self.tape[w][t]
                o = t & 1
                flags = (_lib.TAP_T_FRESH if step == 0 else 0) | (_lib.TAP_T_RATIO if step == self.n - 1 else 0)
                if self.fused and self.bits and t == 0:
                    # a trainer sees fresh fp32 instances every batch (PACKDataset -> DataLoader): the window's
                    # first step reads the fp32 tensor and builds the bit shadow inside the launch
                    self._k("transition_first", L.tap_transition_first, self.ctx, d, P(e._state), self.nw, self.R,
                            self.rows, 3, P(dyn_in), P(st), st.shape[1], P(ptr), P(mask_in), P(self.bitb[o]),
                            P(self.dyn[o]), P(self.cur), P(self.maskb[o]), P(self.feat), P(self.reward),
                            P(self.nonbin), flags)
                    bits_in = self.bitb[o]
                elif self.fused and self.bits:
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

    # ---- after the timed region: the state the LAST pass left behind, against the oracle ---------------
    def verify(self, nenv=VERIFY_ENVS):
        O = _oracle()
        V = min(nenv, self.B)
        self.env.check()
        if self.bits and int(self.nonbin.item()) != 0:
            return dict(verified=False, why="tap_dyn_bits counted non-binary elements")
        st = [s[:V].cpu().numpy() for s in self.static]
        tp = [t[:, :V].t().cpu().numpy() for t in self.tape]       # (V, nw)
        ar = np.arange(V)
        blocks = np.concatenate([np.stack([s[ar, 1:, t[:, k]] for k in range(self.nw)], axis=1)
                                 for s, t in zip(st, tp)], axis=1).astype(np.int32)
        ref = O.run_episodes(O.make_desc(self.cs, self.n, self.reward_type, "diff", self.strategy), blocks,
                             nthreads=_usable_cpus(), want_heightmaps=True)
        if ref["nerr"]:
            return dict(verified=False, why="the oracle flags %d envs of the slice" % ref["nerr"])
        bad = []
        if not np.array_equal(self.reward[:V].cpu().numpy(), ref["ratio"].astype(np.float32)):
            bad.append("calc_ratio")
        if not np.array_equal(self.env.positions[:V].cpu().numpy(), ref["positions"]):
            bad.append("positions")
        if not np.array_equal(self.env.stable[:V].cpu().numpy().astype(np.uint8), ref["stable"]):
            bad.append("stable")
        if not np.array_equal(self.env.heightmap[:V].cpu().numpy().reshape(V, -1), ref["heightmaps"][:, -1]):
            bad.append("heightmap")
        if not np.array_equal(self.feat[:V].cpu().numpy().reshape(V, -1).astype(np.int64), ref["features"][:, -1]):
            bad.append("feature")
        # precedence tensors of the last window after its last step
        dyn = self.dynamic0[-1][:V].cpu().numpy()
        mask = np.ones((V, self.nR), np.float32)
        cur = None
        for k in range(self.nw):
            dyn = O.update_dynamic(dyn, st[-1], tp[-1][:, k], self.nw, 3)
            cur, mask = O.update_mask(mask, dyn, tp[-1][:, k], self.nw, self.R)
        o = (self.nw - 1) & 1
        if not np.array_equal(self.dyn[o][:V].cpu().numpy(), dyn):
            bad.append("dynamic")
        if not np.array_equal(self.cur[:V].cpu().numpy(), cur) or not np.array_equal(self.maskb[o][:V].cpu().numpy(), mask):
            bad.append("masks")
        out = dict(verified=not bad, envs_checked=V,
                   what="last replayed pass vs oracle: calc_ratio (fp32, exact), positions, stable, height-map, last "
                        "feature, final dynamic tensor and both masks")
        if bad:
            out["mismatch"] = bad
        return out


class EpisodeHotPath(HotPath):
    """k6: pack.reward / tools.calc_positions_lb_greedy (pack.py:378-473, tools.py:2393-2449) -- the whole
    episode of every env in ONE launch (tap_episode_reward), the generators' and the reward seam's shape."""

    kind = "episode"

    def __init__(self, cfg, B, start, device, seed=12345):
        HotPath.__init__(self, cfg, B, start, device, seed=seed, instances="generate")
        self.tour = self.tape[0].t().contiguous()             # (B, n)
        self.pos = torch.empty(B, self.n, self.D, dtype=torch.int32, device=device)
        self.stab = torch.empty(B, self.n, dtype=torch.uint8, device=device)
        self.desc = _lib.make_desc(B, self.cs, self.n, self.reward_type, "full", "LB_GREEDY")

    def launches_per_pass(self):
        return 1

    def episode(self):
        P, st = _lib.ptr, self.static[0]
        self._k("episode", self.lib.tap_episode_reward, self.ctx, C.byref(self.desc), self.B, self.n, P(st),
                st.shape[1], self.nR, P(self.tour), P(self.reward), P(self.pos), P(self.stab))

    def verify(self, nenv=VERIFY_ENVS):
        O = _oracle()
        V = min(nenv, self.B)
        st, tour = self.static[0][:V].cpu().numpy(), self.tour[:V].cpu().numpy()
        nerr, want = O.reward(st, tour, self.reward_type, self.cs[0], self.cs[-1], nthreads=_usable_cpus())
        ar = np.arange(V)
        blocks = np.stack([st[ar, 1:, tour[:, k]] for k in range(self.n)], axis=1).astype(np.int32)
        ref = O.run_episodes(O.make_desc(self.cs, self.n, self.reward_type, "full", "LB_GREEDY"), blocks,
                             nthreads=_usable_cpus(), want_heightmaps=False)
        bad = []
        if nerr or not np.array_equal(self.reward[:V].cpu().numpy(), want):
            bad.append("reward")
        if not np.array_equal(self.pos[:V].cpu().numpy(), ref["positions"]):
            bad.append("positions")
        if not np.array_equal(self.stab[:V].cpu().numpy(), ref["stable"]):
            bad.append("stable")
        out = dict(verified=not bad, envs_checked=V, what="last pass vs oracle: -(C+P+S) fp32 exact, positions, stable")
        if bad:
            out["mismatch"] = bad
        return out


class RollingHotPath(HotPath):
    """c5: true rolling windows over device-generated 50-block instances (initial container
    7x7x250 as scripts/rolling.sh); actions replayed from a tape recorded with a random feasible
    policy."""

    rolling = True
    kind = "rolling"

    def __init__(self, cfg, B, start, device, seed=12345, window=10, fused_rolling=False, overlap=False, mix=True):
        _, D, cs, n, _, reward, strategy = cfg
        self.cfg = cfg
        self.overlap = overlap
        self.side = torch.cuda.Stream(device=device)
        self.D, self.cs, self.n, self.B, self.device, self.nw = D, cs, n, B, device, window
        self.reward_type, self.strategy = reward, strategy
        self.fused, self.hook, self.fused_rolling = True, None, fused_rolling
        self.lib, self.ctx = _lib.lib(), _lib.ctx(device)
        self.init = [7, 250] if D == 2 else [7, 7, 250]
        self.mix = bool(mix and hasattr(T.generate, "generate_mix_instances"))
        if self.mix:
            blocks, positions = T.generate.generate_mix_instances(B, n, D, self.init[0], self.init[-1], seed=seed + start,
                                                                   start=start, device=device)
        else:
            _, _, blocks, positions = T.generate.generate_instances(B, n, D, self.init[0], self.init[-1], 1, (1, 5),
                                                                    seed=seed + start, device=device, return_aux=True)
        self.blocks_h, self.positions_h = blocks.cpu().numpy(), positions.cpu().numpy()
        g = torch.Generator(device=device)
        g.manual_seed(seed + 1 + start)
        rec = T.run_rolling_episode(blocks, positions, self.init,
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
        self.want = rec["reward"].clone()                        # -calc_ratio of the eager, unfused recording run

    def launches_per_pass(self):
        per = 1 if self.fused_rolling else 2
        return 2 + (self.n - self.nw) * per + self.nw

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
                P(self.bitb[2] if n1 == 0 else None), P(self.cur), None, None)
        prev_ev = None
        for t in range(n1):
            # the bit shadow is a by-product only the LAST window's full episode consumes
            wbits = self.bitb[2] if t == n1 - 1 else None
            if self.fused_rolling:                                # placement t + window t+1: one launch
                self._k("rolling_step", L.tap_rolling_step, self.ctx, d, P(e._state), self.n, self.nw, P(rw.blocks),
                        P(rw.rel), P(self.state), P(self.tape[t]), P(st[t & 1]), P(st[(t + 1) & 1]), P(self.dyn[2]),
                        P(cs2), P(wbits), P(self.cur), None, None, P(self.feat))
            elif not self.overlap:
                self._k("env_step", L.tap_env_step_gather, self.ctx, d, P(e._state), P(st[t & 1]),
                        self.static.shape[1], self.nR, P(self.tape[t]), None, P(self.feat))
                self._k("rolling_window", L.tap_rolling_window, self.ctx, self.B, self.D, self.n, self.nw,
                        P(rw.blocks), P(rw.rel), P(self.state), P(self.tape[t]), P(st[(t + 1) & 1]), P(self.dyn[2]),
                        P(cs2), P(wbits), P(self.cur), None, None)
            else:
                # placement t and window t+1 on two HIP streams (measured slower, kept as an option)
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
                        P(cs2), P(wbits), P(self.cur), None, None)
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

    def verify(self, nenv=VERIFY_ENVS):
        """(1) every env's calc_ratio against the eager two-launch recording run; (2) a slice against the
        oracle's InitialContainer + Container driven by the same tape: reward, positions, last masks."""
        O = _oracle()
        self.env.check()
        bad = []
        if not torch.equal(self.reward, -self.want):
            bad.append("calc_ratio vs the eager recording run (%d envs differ)" % int((self.reward != -self.want).sum().item()))
        V = min(nenv, self.B)
        tape = self.tape[:, :V].cpu().numpy()                     # (n, V)
        n, win, R = self.n, self.nw, self.R
        got_r = self.reward[:V].cpu().numpy()
        got_pos = self.env.positions[:V].cpu().numpy()
        got_cur = self.cur[:V].cpu().numpy()
        got_dyn = self.dyn[(win - 1) & 1][:V].cpu().numpy()
        nbad = 0
        for b in range(V):
            ro = O.Rolling(self.blocks_h[b], self.positions_h[b], self.init, win)
            e = O.Env(self.cs, n, self.reward_type, "diff", self.strategy)
            for t in range(n - win):
                rc, st, dy, _ = ro.convert_to_input()
                p = int(tape[t, b])
                e.add_new_block(st[1:, p])
                ro.remove(p % win)
            rc, st, dy, _ = ro.convert_to_input()
            mask, dyn, cur = np.ones((1, win * R), np.float32), dy[None], None
            for t in range(win):
                p = np.array([tape[n - win + t, b]], dtype=np.int64)
                e.add_new_block(st[1:, int(p[0])])
                dyn = O.update_dynamic(dyn, st[None], p, win, 3)
                cur, mask = O.update_mask(mask, dyn, p, win, R)
            ok = (np.float32(e.calc_ratio()) == got_r[b] and np.array_equal(e.positions, got_pos[b]) and
                  np.array_equal(cur[0], got_cur[b]) and np.array_equal(dyn[0], got_dyn[b]) and e.error == 0)
            nbad += not ok
        if nbad:
            bad.append("%d of %d envs differ from the oracle" % (nbad, V))
        out = dict(verified=not bad, envs_checked=V, envs_checked_vs_eager=self.B,
                   what="last replayed pass: calc_ratio of all envs vs the eager unfused run; slice vs the oracle "
                        "(InitialContainer windows + Container, same tape): calc_ratio, positions, last dynamic and mask")
        if bad:
            out["mismatch"] = bad
        return out


GATHER_EVERY = 8   # least number of passes per graph launch / reward vectors per RCCL all-gather (passes_per_graph)


def passes_per_graph(hp):
    """Passes replayed per graph launch (= reward vectors per all-gather): 8 for passes of 0.4 ms and more, up to 64
    for the shortest, so that a graph launch (~5 us) and, with N > 1, an all-gather launch (~60 us between two
    graph launches, measured on a 1-rank RCCL group) stay small against the work they bracket.  The estimate is an
    eager pass, timed once."""
    for _ in range(2):
        hp.episode()
    torch.cuda.synchronize(hp.device)
    t0 = time.perf_counter()
    for _ in range(3):
        hp.episode()
    torch.cuda.synchronize(hp.device)
    t = (time.perf_counter() - t0) / 3
    return GATHER_EVERY if t >= 400e-6 else 2 * GATHER_EVERY if t >= 200e-6 else 4 * GATHER_EVERY if t >= 100e-6 else 8 * GATHER_EVERY


def time_passes(hps, steps, warmup, use_graph, world, repeats=1):
    """Time `steps` passes, pass i on slot i % len(hps) (one slot = the headline; several = the cold variant)."""
    if not isinstance(hps, (list, tuple)):
        hps = [hps]
    hp = hps[0]
    dev = hp.device
    handles = []
    import torch.distributed as tdd
    # TAP_BENCH_FORCE_GATHER=1 runs the N > 1 code path (accumulate + async all-gather) in a 1-rank
    # process group: the only way to time its overhead on a 1-GPU box
    gather = world > 1 or (os.environ.get("TAP_BENCH_FORCE_GATHER") == "1" and tdd.is_available() and tdd.is_initialized())
    nranks = tdd.get_world_size() if gather else 1
    # With a gather, pass i writes its (B,) reward vector straight into row i % ge of `acc` (the pass's
    # ratio_out IS that row) -- no copy kernel (measured: ~15 us per pass as an eager launch between two replays,
    # ~7 us as an extra graph node), and ge passes share one all-gather.
    single = len(hps) == 1
    ge = passes_per_graph(hp) if single else GATHER_EVERY
    hp.passes_per_graph = ge
    # (a single instance set files its passes' rewards into the rows without a gather too: the rows of the last
    #  group are compared with each other afterwards -- the tape is the same every pass, so they must be identical;
    #  this is what caught a hipGraph memset node running out of order in the rolling pass)
    rows_on = gather or single
    acc = torch.full((ge, hp.B), float("nan"), dtype=torch.float32, device=dev) if rows_on else None
    if gather and not single:
        raise ValueError("the gather path times one instance set")
    state = {"i": 0, "pass": 0}

    gflat = torch.empty(nranks * ge * hp.B, dtype=torch.float32, device=dev) if gather else None
    last_gather = {}

    def flush(count):
        # the only exchange of the whole job: the group's rows, straight from `acc` into one preallocated tensor
        # (stream-ordered call: the next graph launch waits for it, so the rows may be overwritten).  A list-output
        # async all_gather of a cloned buffer cost 10 % at c2 on a 1-rank RCCL group, this form 2-4 %.
        import torch.distributed as dist
        out = gflat[: nranks * count * hp.B].view(nranks * count, hp.B)
        dist.all_gather_into_tensor(out, acc[:count])
        last_gather["out"], last_gather["count"] = out, count

    def drain():
        for h, _ in handles:
            h.wait()
        handles.clear()

    def warm(h):
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            h.episode()                                         # warm the allocator / lazy init
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)

    def capture(h, npasses):
        """ONE graph of `npasses` consecutive passes of h (with a gather: rows 0 .. npasses-1 of acc)."""
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for j in range(npasses):
                if rows_on:
                    h.reward = acc[j]
                h.episode()
        return g

    graphs = None
    group = {}
    if use_graph:
        for h in hps:
            warm(h)
        graphs = [capture(h, 1) for h in hps]                   # one pass each: event timing, the cold rotation
        if single:
            # the timed loop of one instance set replays ge passes per graph launch (and one shorter
            # graph for the remainder): the launch-bound inner loop is what a hipGraph is for
            for cnt in sorted({ge, warmup % ge, steps % ge} - {0}):
                group[cnt] = capture(hp, cnt)
        if rows_on:
            hp.reward = acc[0]                                  # graphs[0] (event timing, verification) writes row 0
        if single and ge in group:
            hp.group_graph = (group[ge], ge)
        if not single:
            # the rotation over several instance sets (the cold reading) as ONE graph of one pass per set: the same
            # launch amortisation as the headline's groups.  (Up to round 6 it was one graph launch per PASS, whose
            # ~8 us of launch gap per 10 kernels read as part of the cold penalty: 77.8 against 69.7 us per pass at c2,
            # the second figure from scripts/decompose_step.py's single-graph rotation.)
            cycle = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cycle):
                for h in hps:
                    h.episode()
            group["cycle"] = cycle

    def run(npasses):
        if single:
            done = 0
            while done < npasses:
                cnt = min(ge, npasses - done)
                if graphs is not None:
                    group[cnt].replay()
                else:
                    for j in range(cnt):
                        hp.reward = acc[j]
                        hp.episode()
                if gather:
                    flush(cnt)
                done += cnt
        else:
            done = 0
            while done < npasses:
                k = state["pass"] % len(hps)
                if graphs is not None and k == 0 and npasses - done >= len(hps):
                    group["cycle"].replay()                     # one pass on every set
                    state["pass"] += len(hps)
                    done += len(hps)
                    continue
                state["pass"] += 1
                done += 1
                if graphs is not None:
                    graphs[k].replay()
                else:
                    hps[k].episode()

    first_gather_ms = None
    if gather:
        # RCCL sets its rings up lazily inside the first collective: pay (and report) that before the warm-up,
        # never inside the timed bracket
        torch.cuda.synchronize(dev)
        tg = time.perf_counter()
        flush(1)
        torch.cuda.synchronize(dev)
        first_gather_ms = (time.perf_counter() - tg) * 1e3
    hp.first_gather_ms = first_gather_ms
    run(warmup)
    drain()
    dts, local_dts = [], []
    for _ in range(max(1, repeats)):
        # one bracket = EXACTLY `steps` passes between barrier + synchronize on both sides, MAX over ranks.  A rank's clock
        # stops when ITS work (exchange included) is complete, the closing barrier follows and the job's time is the MAX
        # over the ranks: the barrier's own latency (an RCCL all-reduce) is not part of any rank's `steps` passes, and the
        # per-rank figures in `ranks.per_rank` stay each rank's own.
        tdist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        run(steps)
        drain()
        torch.cuda.synchronize(dev)
        local_dt = time.perf_counter() - t0
        tdist.barrier()
        local_dts.append(local_dt)
        dts.append(tdist.max_over_ranks(local_dt, dev))
    hp.bracket_times = dts
    hp.local_bracket_times = local_dts
    dt = statistics.median(dts)
    if gather and last_gather:
        r, c = tdd.get_rank(), last_gather["count"]
        mine = last_gather["out"][r * c:(r + 1) * c]
        hp.gathered_ok = bool(torch.equal(torch.nan_to_num(mine, nan=-7.0), torch.nan_to_num(acc[:c], nan=-7.0)))
    if rows_on:
        hp.reward = acc[0]
        # every pass of the last group(s) against row 0 (NaN = a row no timed pass wrote; NaN rewards of flagged
        # containers compare equal to themselves here)
        written = [j for j in range(ge) if j < min(steps, ge)]
        same = all(bool(torch.equal(torch.nan_to_num(acc[j], nan=-7.0), torch.nan_to_num(acc[0], nan=-7.0))) for j in written)
        hp.passes_identical = dict(passes_compared=len(written), identical=same)
    return dt, graphs


def kernel_event_times(hp, steps, graph=None):
    """Per-launch durations with HIP events on the launch stream (torch.cuda.Event on torch's
    current stream == the stream the kernels are enqueued on), over `steps` eager passes.  Every launch
    is bracketed by its own event pair; the cost of an empty pair is measured next to it."""
    recs = {}

    def hook(name, launch):
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        launch()
        b.record()
        recs.setdefault(name, []).append((a, b))

    # second form: ONE event pair around each run of consecutive launches of the same kernel (the 40 rolling steps of
    # a pass ...): elapsed / launches = per-launch time incl. the gap between launches, with the event overhead
    # spread over the run -- a single launch between two events reads 2-4 us long, and subtracting an empty pair
    # (5-7 us) over-corrects
    runs, cur = {}, {"name": None, "start": None, "n": 0}

    def close_run():
        if cur["name"] is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            runs.setdefault(cur["name"], []).append((cur["start"], e, cur["n"]))
        cur["name"], cur["start"], cur["n"] = None, None, 0

    def run_hook(name, launch):
        if name != cur["name"]:
            close_run()
            cur["name"] = name
            cur["start"] = torch.cuda.Event(enable_timing=True)
            cur["start"].record()
        launch()
        cur["n"] += 1

    pass_pairs = []
    hp.hook = hook
    try:
        for _ in range(steps):
            hp.episode()
        torch.cuda.synchronize(hp.device)
        hp.hook = run_hook
        for _ in range(steps):
            hp.episode()
            close_run()
        torch.cuda.synchronize(hp.device)
        # whole passes without per-launch events: (pass time) / launches = per-launch time incl. the
        # inter-kernel gap, free of the event-pair overhead
        hp.hook = None
        grp = getattr(hp, "group_graph", None) if graph is not None else None   # (graph of cnt passes, cnt): the timed loop's unit
        per = grp[1] if grp else 1
        # (at least seven replays, the first one dropped below: with two, the figure was the mean of a first replay that
        #  follows a burst of eager launches and one steady replay -- 6.09 against 5.46 us per launch at c2 between two runs
        #  of the same build, round 6)
        for _ in range(max(7, steps // per)):
            a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
            a.record()
            if grp:
                grp[0].replay()
            elif graph is not None:
                graph.replay()
            else:
                hp.episode()
            b.record()
            pass_pairs.append((a, b))
        torch.cuda.synchronize(hp.device)
    finally:
        hp.hook = None
    pairs = []
    for _ in range(64):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record(); b.record(); pairs.append((a, b))
    torch.cuda.synchronize(hp.device)
    empty_us = float(np.median([a.elapsed_time(b) for a, b in pairs]) * 1e3)
    out = {}
    for name, evs in recs.items():
        us = np.array([a.elapsed_time(b) for a, b in evs]) * 1e3
        out[name] = dict(launches=len(us), avg_us=float(us.mean()), med_us=float(np.median(us)),
                         total_us=float(us.sum()))
        rr = runs.get(name, [])
        if rr:
            per_launch_us = np.array([a.elapsed_time(b) * 1e3 / n for a, b, n in rr])
            out[name]["run_us"] = float(np.median(per_launch_us))
            out[name]["run_len"] = int(np.median([n for _, _, n in rr]))
    pass_us = float(np.median([a.elapsed_time(b) for a, b in pass_pairs[1:]]) * 1e3) / per
    return out, empty_us, pass_us


def cpu_info():
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    try:                                     # a container's CPU quota (cgroup v2) caps what threads can use
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            avail = max(1, min(avail, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return model, os.cpu_count() or 1, avail


def _timed_loop(fn, budget_s):
    done, t0 = 0, time.perf_counter()
    while True:
        done += fn()
        el = time.perf_counter() - t0
        if el > budget_s:
            return done, el


def cpu_baseline(hp, budget_s=10.0):
    """The oracle (C port of the reference algorithm) over the same pass -- masks + placements + ratio of the
    first <= 4096 instances of THIS run -- on one host core, then on all of them (OpenMP over envs)."""
    O = _oracle()
    B = min(hp.B, 4096)
    nw, n = hp.nw, hp.n
    wins = [(s[:B].cpu().numpy(), d[:B].cpu().numpy(), t[:, :B].t().cpu().numpy())
            for s, d, t in zip(hp.static, hp.dynamic0, hp.tape)]
    R = hp.R
    ar = np.arange(B)
    blocks = np.concatenate([np.stack([st[ar, 1:, tape[:, t]] for t in range(nw)], axis=1)
                             for st, _, tape in wins], axis=1).astype(np.int32)
    desc = O.make_desc(hp.cs, n, hp.reward_type, "diff", hp.strategy)
    O.lib()

    def one(threads):
        if hp.kind == "episode":
            nerr, _ = O.reward(wins[0][0], wins[0][2], hp.reward_type, hp.cs[0], hp.cs[-1], nthreads=threads)
            assert nerr == 0
            return B * n
        for st, dyn0, tape in wins:
            dyn, mask = dyn0, np.ones((B, st.shape[2]), np.float32)
            O.initial_mask(dyn, nw)
            for t in range(nw):
                dyn = O.update_dynamic(dyn, st, tape[:, t], nw, 3)
                _, mask = O.update_mask(mask, dyn, tape[:, t], nw, R)
        r = O.run_episodes(desc, blocks, nthreads=threads, want_heightmaps=False)
        assert r["nerr"] == 0
        return B * n

    model, ncpu, avail = cpu_info()
    O.set_threads(1)
    done1, el1 = _timed_loop(lambda: one(1), budget_s)
    O.set_threads(avail)
    doneN, elN = _timed_loop(lambda: one(avail), budget_s / 2)
    O.set_threads(1)
    what = "whole episodes (pack.reward)" if hp.kind == "episode" else "masks + placements + ratio"
    return dict(value=done1 / el1, unit="env-steps/s", cores=1, kind="port",
                sample="%d passes of B=%d envs x n=%d (%s) of this run's instances, %.1f s, "
                       "oracle/libtap_oracle.so single thread" % (done1 // (B * n), B, n, what, el1),
                all_cores=dict(value=doneN / elN, cores=avail,
                               sample="%d passes, %.1f s, OpenMP over envs, %d threads (CPUs this process may use: "
                                      "affinity and cgroup quota)" % (doneN // (B * n), elN, avail)),
                cpu_model=model, cpu_count=ncpu, cpus_usable=avail)


def cpu_baseline_rolling(hp, budget_s=10.0):
    """The oracle over rolling.validate's loop (windows re-cut after every placement) on this run's instances
    and tape: one host core (driven per step from Python), then one process per core."""
    O = _oracle()
    n, win, R = hp.n, hp.nw, hp.R
    V = min(hp.B, 256)
    tape = hp.tape[:, :V].cpu().numpy()
    O.lib()

    def episodes(lo, hi):
        for b in range(lo, hi):
            ro = O.Rolling(hp.blocks_h[b], hp.positions_h[b], hp.init, win)
            e = O.Env(hp.cs, n, hp.reward_type, "diff", hp.strategy)
            for t in range(n - win):
                rc, st, dy, _ = ro.convert_to_input()
                O.initial_mask(dy[None], win)
                p = int(tape[t, b])
                e.add_new_block(st[1:, p])
                ro.remove(p % win)
            rc, st, dy, _ = ro.convert_to_input()
            mask, dyn = np.ones((1, win * R), np.float32), dy[None]
            for t in range(win):
                p = np.array([tape[n - win + t, b]], dtype=np.int64)
                e.add_new_block(st[1:, int(p[0])])
                dyn = O.update_dynamic(dyn, st[None], p, win, 3)
                _, mask = O.update_mask(mask, dyn, p, win, R)
            e.calc_ratio()
        return (hi - lo) * n

    model, ncpu, avail = cpu_info()
    O.set_threads(1)
    done1, el1 = _timed_loop(lambda: episodes(0, V), budget_s)
    # all cores: the per-step driver is Python, so one forked process per usable CPU, each looping over its own
    # slice of the instances for the same wall time
    import multiprocessing as mp
    ctxm = mp.get_context("fork")
    per = max(1, V // avail)
    q = ctxm.Queue()

    def worker(lo, hi, seconds):
        d, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            d += episodes(lo, hi)
        q.put(d)

    t0 = time.perf_counter()
    procs = []
    for k in range(avail):
        lo, hi = k * per, min(V, (k + 1) * per)
        if lo < hi:
            pr = ctxm.Process(target=worker, args=(lo, hi, budget_s / 2))
            pr.start()
            procs.append(pr)
    doneN = sum(q.get() for _ in procs)
    for pr in procs:
        pr.join()
    elN = time.perf_counter() - t0
    return dict(value=done1 / el1, unit="env-steps/s", cores=1, kind="port",
                sample="%d rolling episodes of %d placements (window %d) over this run's instances and tape, %.1f s, "
                       "oracle/libtap_oracle.so single thread driven per step from Python" % (done1 // n, n, win, el1),
                all_cores=dict(value=doneN / elN, cores=len(procs),
                               sample="%d episodes, %.1f s, one forked process per usable CPU, each looping over its "
                                      "slice of the instances" % (doneN // n, elN)),
                cpu_model=model, cpu_count=ncpu, cpus_usable=avail)


def attach_reference_cpu(cb, config):
    """The reference's own Python path cannot run on the GPU box; scripts/time_reference.py measured it in the
    build container (1 process and 8 processes) next to the oracle on the same work.  First-class fields:
    reference_value = this box's oracle figure / that ratio (derived), reference_measured = the build
    container's own figure (measured there)."""
    for name in ("r03_reference_cpu.jsonl", "r02_reference_cpu.jsonl", "r01d_reference_cpu.jsonl"):
        rp = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(rp):
            continue
        key = {"c2": "c1/c2", "k6": "c1/c2", "c3": "c3", "c4": "c4", "c5": "c5"}.get(config)
        for line in open(rp):
            r = json.loads(line)
            if r.get("config") != key or config == "k6":
                continue
            ratio = r["ratio_oracle_over_reference"]
            cb["reference_value"] = cb["value"] / ratio
            cb["reference_cores"] = 1
            cb["reference_how"] = ("derived: this box's 1-core oracle figure / %.0f (reference vs oracle on one core of the "
                                   "build container, profiles/%s)" % (ratio, name))
            cb["reference_measured"] = dict(
                where="build container (the only place /root/reference exists)",
                one_process=r.get("reference_env_steps_per_s"),
                all_processes=r.get("reference_env_steps_per_s_procs"), processes=r.get("procs"),
                cpu_model=r.get("cpu_model"), source="profiles/" + name)
            return


def load_traffic(key):
    """HBM bytes per launch from the PMC passes of the profile set named in the entry (profiles/README.md:
    separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, (2*FETCH + WRITE)*1024)."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            t = json.load(open(p)).get(key)
        except Exception:
            return None
        if isinstance(t, dict):
            return t
        if t is not None:
            return dict(bytes=t, profile="r01d (round 1)")
    return None


def pass_compulsory_bytes(hp, cfg, rolling):
    """Compulsory bytes of ONE whole pass of hp (every launch of it, compulsory_bytes per kind) for all its envs."""
    name, D, cs, n, B, reward, strategy = cfg
    bits_on = bool(getattr(hp, "bits", True)) or rolling
    cb = lambda k: compulsory_bytes(k, D, cs, hp.nw, bits=bits_on) * B     # noqa: E731
    if rolling:
        return cb("rolling_window") + (n - hp.nw) * cb("rolling_step") + hp.nw * cb("transition")
    if hp.kind == "episode":
        return episode_bytes(D, n) * B
    if bits_on:
        return hp.windows * (cb("transition_first") + (hp.nw - 1) * cb("transition"))
    return n * cb("transition")


def cold_reading(cfg, hp, rolling, dev, use_graph, max_slots=40, min_steps=40):
    """The same pass with nothing cache-resident: pass i runs on instance set i % slots, each set with its own input
    AND output buffers, the sets together several times the 256 MB Infinity Cache.  A trainer with a network between
    the steps lives nearer this figure than the headline's (one instance set replayed, ~65 MB at c2)."""
    name, D, cs, n, B, reward, strategy = cfg
    if rolling:
        per_slot = sum(t.numel() * t.element_size() for t in ([hp.rw.rel, hp.rw.blocks, hp.static, hp.static2, hp.cur, hp.feat]
                                                              + hp.dyn + hp.maskb + [b for b in hp.bitb if b is not None]))
        slots = min(12, max(3, int(np.ceil(1.2e9 / per_slot))), max_slots)
        hps = [hp] + [RollingHotPath(cfg, B, 0, dev, seed=777 + 1000 * k, window=hp.nw, fused_rolling=hp.fused_rolling, mix=hp.mix)
                      for k in range(1, slots)]
        steps = slots * 3
    else:
        per_slot = sum(t.numel() * t.element_size() for t in (hp.dynamic0 + hp.static + hp.dyn + [hp.cur] + hp.maskb))
        slots = min(max(3, int(np.ceil(1.2e9 / per_slot))), max_slots)

        def slot_instances(k):
            if hp.instances is None or isinstance(hp.instances, str) and hp.instances != "ppsg2d":
                return hp.instances
            # PPSG instances (30 s per 8 192 from the device generator; 64 in the fixture): the other slots hold this
            # run's instances in another order (own buffers, own tape) -- the reading is about residency, not diversity
            return (hp.static[0].roll(977 * k, 0).cpu().numpy(), hp.dynamic0[0].roll(977 * k, 0).cpu().numpy())
        hps = [hp] + [HotPath(cfg, B, 0, dev, seed=777 + 1000 * k, fused=hp.fused, window=hp.nw if hp.windows > 1 else None,
                              bits=hp.bits, instances=slot_instances(k)) for k in range(1, slots)]
        steps = slots * max(4, -(-min_steps // slots))       # whole rotations: every timed pass sits in a cycle graph
    dt, _ = time_passes(hps, steps, slots, use_graph, 1)
    for h in hps:
        h.env.check()
    pass_us = dt / steps * 1e6
    ach = pass_compulsory_bytes(hp, cfg, rolling) / (pass_us * 1e-6) / 1e9
    out = dict(value=B * n * steps / dt, unit="env-steps/s", pass_us=pass_us, achieved=ach, frac=ach / HBM_PEAK_GBS,
               frac_how="compulsory bytes of every launch of a pass / pass time / 8 TB/s (gaps between the launches included)",
               slots=slots, working_set_MB=round(per_slot * slots / 1e6, 1), steps=steps,
               what="pass i runs on instance set i %% %d, each with its own input and output buffers; the working set is "
                    "several times the 256 MB Infinity Cache; one graph launch per rotation over the sets" % slots)
    del hps
    torch.cuda.empty_cache()
    return out


def variants_rolling(cfg, hp, dev, use_graph, trace):
    """c5's two other readings: (a) cold -- the passes rotate over enough instance sets (relation masks, blocks,
    window and container buffers per set) to exceed the Infinity Cache; (b) rolling.validate's loop with a policy
    between the steps: run_rolling_episode, RandomFeasiblePolicy, eager launches."""
    name, D, cs, n, B, reward, strategy = cfg
    out = {}
    skip = os.environ.get("TAP_BENCH_SKIP", "").split(",")
    trace("cold")
    try:
        if "cold" in skip:
            raise RuntimeError("skipped")
        out["cold"] = getattr(hp, "cold", None) or cold_reading(cfg, hp, True, dev, use_graph)
    except Exception as ex:                                  # pragma: no cover
        out["cold"] = dict(error=str(ex))
    trace("policy_in_loop")
    try:
        if "eager" in skip:
            raise RuntimeError("skipped")
        g = torch.Generator(device=dev)
        g.manual_seed(4242)
        blocks = hp.rw.blocks
        positions = torch.as_tensor(hp.positions_h, device=dev)
        pair = [None]

        def run(pol):
            r = T.run_rolling_episode(blocks, positions, hp.init, pol, cs[0], cs[-1], child_graph_size=hp.nw,
                                      reward_type=reward, steppers=pair[0])
            pair[0] = r["steppers"]
            return r

        def eager(pol, steps):
            run(pol)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                r = run(pol)
            t1 = time.perf_counter()
            torch.cuda.synchronize(dev)
            t2 = time.perf_counter()
            r["env"].check(); r["windows"].check()
            return B * n * steps / (t2 - t0), (t1 - t0) / (steps * n) * 1e6, r

        class FirstSelectable(object):                           # null policy: the first selectable column, one torch op
            def __call__(self, current_mask, **_):
                return torch.argmax(current_mask, dim=1)
        v0, host0, _ = eager(FirstSelectable(), 5)
        out["host_us_per_step"] = dict(value=host0, unit="us", eager_env_steps_per_s=v0,
                                       what="host time to ISSUE one decoding step of rolling.run_rolling_episode on a persistent "
                                            "RollingStepper / EpisodeStepper pair with a one-op policy (argmax of current_mask): "
                                            "one C call (tap_roller_step / tap_stepper_step) + the launch + the policy's op")
        v1, host1, _ = eager(T.RandomFeasiblePolicy(g), 5)
        out["policy_in_loop"] = dict(value=v1, unit="env-steps/s", steps=5, host_us_per_step=host1,
                                     what="rollout.run_rolling_episode on a persistent RollingStepper / EpisodeStepper pair "
                                          "(relation masks rebuilt per episode, one fused tap_rolling_step per window, then the "
                                          "last window's episode), RandomFeasiblePolicy (exponential race on current_mask, 3 "
                                          "torch ops) between the steps, eager launches")
    except Exception as ex:                                  # pragma: no cover
        out["policy_in_loop"] = dict(error=str(ex)[:300])
    trace("no_fp32_expand")
    try:
        if "eager" in skip:
            raise RuntimeError("skipped")
        # the windows' fp32 precedence tensor (7 200 of the 9 937 bytes a c5 step moves) left out: the same episodes,
        # this run's recorded tour replayed, on step objects with and without the expansion (tapenv.h: dynamic = NULL)
        blocks = hp.rw.blocks
        positions = torch.as_tensor(hp.positions_h, device=dev)
        tape = T.TapePolicy(hp.tape.t().contiguous())                 # (B, n): the recorded tour
        pairs = {True: None, False: None}

        def run2(expand):
            r = T.run_rolling_episode(blocks, positions, hp.init, tape, cs[0], cs[-1], child_graph_size=hp.nw,
                                      reward_type=reward, steppers=pairs[expand], expand_dynamic=expand)
            pairs[expand] = r["steppers"]
            return r

        def timed2(expand, steps):
            run2(expand)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                r = run2(expand)
            torch.cuda.synchronize(dev)
            return B * n * steps / (time.perf_counter() - t0), r
        v_on, r_on = timed2(True, 5)
        keep = [r_on[k].clone() for k in ("reward", "tour_idx", "nodes")]
        v_off, r_off = timed2(False, 5)
        same = all(torch.equal(torch.nan_to_num(a), torch.nan_to_num(b)) for a, b in zip(keep, [r_off[k] for k in ("reward", "tour_idx", "nodes")]))
        out["no_fp32_expand"] = dict(value=v_off, unit="env-steps/s", with_expansion=v_on, steps=5, identical_outputs=bool(same),
                                     dynamic_is_none=pairs[False][0].dynamic is None,
                                     what="rolling.run_rolling_episode(expand_dynamic=False) against expand_dynamic=True on persistent "
                                          "step objects, this run's tour replayed (TapePolicy), eager launches: the windows' fp32 "
                                          "precedence tensors are never written, a policy reads their bit shadows")
    except Exception as ex:                                  # pragma: no cover
        out["no_fp32_expand"] = dict(error=str(ex)[:300])
    return out


class ActorShapedPolicy(torch.nn.Module):
    """A stand-in policy with the op count and tensor shapes of the reference's actor at inference (model.py:141-515: 1x1
    convolutions as encoders of `static`, of the per-step `dynamic` and of the decoder inputs, one GRU step, the
    attention over the columns, the context read, the pointer energies, the mask added as a log, softmax) at the
    reference's hidden size 128 -- random weights, written here from that description, not the reference's module.
    It exists to answer one question: what share of a decoding step is the environment when a network of the actor's
    size sits between the steps.  Sampling: the Gumbel-max draw of the masked softmax from pre-drawn uniform keys
    (keys (steps, B, nR), refreshed outside the graph), so an episode with it can be captured in a hipGraph."""

    def __init__(self, D, rows, nR, flen, keys, hidden=128):
        super().__init__()
        H = hidden
        nn = torch.nn
        self.static_enc = nn.Conv1d(1 + D, H, 1)
        self.dynamic_enc = nn.Conv1d(rows, H, 1)
        self.dec_static = nn.Conv1d(D, H // 2, 1)
        self.dec_dynamic = nn.Linear(flen, H // 2)
        self.gru = nn.GRUCell(H, H)
        self.Wa = nn.Parameter(torch.randn(H, 3 * H) * 0.05)
        self.va = nn.Parameter(torch.randn(1, H) * 0.05)
        self.Wp = nn.Parameter(torch.randn(H, 4 * H) * 0.05)
        self.vp = nn.Parameter(torch.randn(1, H) * 0.05)
        self.keys = keys
        self.hh = None
        self.static_hidden = None
        self.ops_per_step = 24

    @torch.no_grad()
    def begin(self, static, B):
        self.static_hidden = self.static_enc(static)                       # once per episode (model.py:282)
        self.hh = torch.zeros(B, self.gru.hidden_size, device=static.device)

    @torch.no_grad()
    def forward(self, step, dynamic, current_mask, decoder_static, decoder_dynamic, **_):
        B = current_mask.shape[0]
        dh = self.dynamic_enc(dynamic)                                     # (B, H, nR)
        dec = torch.cat((self.dec_static(decoder_static).squeeze(2),
                         self.dec_dynamic(decoder_dynamic.reshape(B, -1))), 1)      # (B, H)
        self.hh = self.gru(dec, self.hh)
        enc = torch.cat((self.static_hidden, dh), 1)                       # (B, 2H, nR)
        hid = torch.cat((enc, self.hh.unsqueeze(2).expand(-1, -1, enc.shape[2])), 1)
        attn = torch.softmax(torch.matmul(self.va, torch.tanh(torch.matmul(self.Wa, hid))), dim=2)    # (B, 1, nR)
        ctx = torch.bmm(attn, enc.transpose(1, 2)).transpose(1, 2).expand_as(enc)                    # (B, 2H, nR)
        logits = torch.matmul(self.vp, torch.tanh(torch.matmul(self.Wp, torch.cat((enc, ctx), 1)))).squeeze(1)
        probs = torch.softmax(logits + current_mask.log(), dim=1)          # model.py:393-395
        # Gumbel-max: argmax(log p - log(-log u)) is a draw from p; masked columns stay at -inf
        return torch.argmax(probs.log() - torch.log(-torch.log(self.keys[step])), dim=1)


def variants(cfg, args, hp, rank, world, dev, use_graph):
    """Two more readings of the same workload (single GPU, rank 0's line only)."""
    name, D, cs, n, B, reward, strategy = cfg
    out = {}
    if hp.kind not in ("transition", "rolling") or world != 1:
        return out
    trace = (lambda m: print("variants: " + m, file=sys.stderr, flush=True)) if os.environ.get("TAP_BENCH_TRACE") else (lambda m: None)
    if hp.kind == "rolling":
        return variants_rolling(cfg, hp, dev, use_graph, trace)
    # (a) cold: rotate over enough instance batches (inputs AND output buffers per slot) that the working set
    #     exceeds the 256 MB Infinity Cache several times over -- nothing a pass reads is cache-resident
    skip = os.environ.get("TAP_BENCH_SKIP", "").split(",")
    trace("cold")
    try:
        if "cold" in skip:
            raise RuntimeError("skipped")
        out["cold"] = getattr(hp, "cold", None) or cold_reading(cfg, hp, False, dev, use_graph)
    except Exception as ex:                                  # pragma: no cover
        out["cold"] = dict(error=str(ex))
    # (b) the loop a trainer runs: its policy between the steps, eager launches, ONE persistent pack.EpisodeStepper
    #     (buffers owned by the stepper, one C call per step; pack.set_binary_check('trust'): no host read per episode)
    st, dy = hp.static[0], hp.dynamic0[0]
    cw, ch = cs[0], cs[-1]
    O = _oracle()

    def oracle_ok(rec):
        V = min(VERIFY_ENVS, B)
        tour = rec["tour_idx"][:V].cpu().numpy()
        stn = st[:V].cpu().numpy()
        blocks = np.stack([stn[np.arange(V), 1:, tour[:, k]] for k in range(hp.nw)], axis=1).astype(np.int32)
        want = O.run_episodes(O.make_desc(cs, hp.nw, reward, "diff", strategy), blocks, nthreads=_usable_cpus(),
                              want_heightmaps=False)
        return bool(want["nerr"] == 0 and np.array_equal(rec["reward"][:V].cpu().numpy(), -want["ratio"].astype(np.float32)))

    def eager(pol, steps, stepper, check=True):
        """-> (env-steps/s incl. drain, host us per decoding step to ISSUE the loop, record of the last episode)"""
        run = lambda: T.run_episode(st, dy, pol, cw, ch, reward_type=reward, packing_strategy=strategy, stepper=stepper)  # noqa: E731
        for _ in range(3):
            r = run()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            r = run()
        t1 = time.perf_counter()
        torch.cuda.synchronize(dev)
        t2 = time.perf_counter()
        if check:
            r["stepper"].check()
            T.pack.check_binary()
        return B * hp.nw * steps / (t2 - t0), (t1 - t0) / (steps * hp.nw) * 1e6, r

    try:
        T.pack.set_binary_check('trust')
        if "eager" in skip:
            raise RuntimeError("skipped")
        trace("policy_in_loop")
        env_v = T.BatchedContainer(B, cs, hp.nw, reward, "diff", packing_strategy=strategy, device=dev)
        sp = T.EpisodeStepper(st, dy, env_v, steps=hp.nw)
        # null policy: a recorded tour -- what the host pays per decoding step for OUR side of the loop
        tape_pol = T.TapePolicy(hp.tape[0].t())                 # hp.tape: (nw, B) per window -> (B, nw)
        v_tape, host_us, r = eager(tape_pol, 100, sp)
        out["host_us_per_step"] = dict(value=host_us, unit="us", eager_env_steps_per_s=v_tape, verified=oracle_ok(r),
                                       what="host time to ISSUE one decoding step of rollout.run_episode on a persistent "
                                            "pack.EpisodeStepper with a null policy (a recorded tour): argument checks + one C "
                                            "call (tap_stepper_step) + the launch; eager, no graph; the env-steps/s figure "
                                            "includes the drain (barrier-to-barrier)")
        g = torch.Generator(device=dev)
        g.manual_seed(4242)
        v_pol, host_pol, r = eager(T.RandomFeasiblePolicy(g), 60, sp)
        out["policy_in_loop"] = dict(value=v_pol, unit="env-steps/s", steps=60, host_us_per_step=host_pol, verified=oracle_ok(r),
                                     what="rollout.run_episode on a persistent pack.EpisodeStepper, RandomFeasiblePolicy "
                                          "(uniformly random selectable column: an exponential race on current_mask, 3 torch "
                                          "ops) between the fused steps, eager launches, bit shadow and initial mask built "
                                          "per episode; pack.set_binary_check('trust')")
        g.manual_seed(4242)
        v_mn, host_mn, r = eager(T.rollout.MultinomialPolicy(g), 30, sp)
        out["policy_in_loop_multinomial"] = dict(value=v_mn, unit="env-steps/s", steps=30, host_us_per_step=host_mn,
                                                 what="the same loop with round 3's stand-in policy, torch.multinomial on "
                                                      "current_mask (~85 us of host per call by itself on this stack)")
    except Exception as ex:                                  # pragma: no cover
        out.setdefault("policy_in_loop", dict(error=str(ex)[:300]))
    # (c) the same loop -- policy included -- captured once in a hipGraph: the seams make no host read, so episode set-up
    #     (fresh container, shadow + initial mask in one launch), the policy's torch ops and the fused steps all replay
    def graphed(pol, refresh, stepper, steps=100):
        run2 = lambda: T.run_episode(st, dy, pol, cw, ch, reward_type=reward, packing_strategy=strategy, stepper=stepper)  # noqa: E731
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            run2(); run2()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            rec = run2()
        for _ in range(3):
            refresh()
            graph.replay()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            refresh()
            graph.replay()
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        rec["stepper"].check()
        T.pack.check_binary()
        return B * hp.nw * steps / dt, rec

    try:
        if "graph" in skip:
            raise RuntimeError("skipped")
        trace("policy_in_loop_graph")
        g2 = torch.Generator(device=dev)
        g2.manual_seed(4243)
        env_g = T.BatchedContainer(B, cs, hp.nw, reward, "diff", packing_strategy=strategy, device=dev)
        spg = T.EpisodeStepper(st, dy, env_g, steps=hp.nw)
        u = torch.rand(B, hp.nw, device=dev, generator=g2)      # refreshed (eagerly) before every replay
        val, rec = graphed(T.UniformPickPolicy(u), lambda: u.uniform_(generator=g2), spg)
        out["policy_in_loop_graph"] = dict(value=val, unit="env-steps/s", steps=100, verified=oracle_ok(rec),
                                           what="rollout.run_episode with UniformPickPolicy (k-th selectable column by "
                                                "cumulative sum, ~10 torch ops per step, from uniforms drawn before each "
                                                "replay) captured in one hipGraph -- shadow + initial mask, the policy's "
                                                "torch ops and the fused steps; the replayed tour is checked against the oracle")
        trace("policy_in_loop_graph_keys")
        keys = torch.empty(hp.nw, B, st.shape[2], device=dev).uniform_(1e-7, 1.0, generator=g2)
        val, rec = graphed(T.UniformKeysPolicy(keys), lambda: keys.uniform_(1e-7, 1.0, generator=g2), spg)
        out["policy_in_loop_graph_keys"] = dict(value=val, unit="env-steps/s", steps=100, verified=oracle_ok(rec),
                                                what="the same with UniformKeysPolicy: argmax of pre-drawn iid keys over the "
                                                     "selectable columns (uniform over them), 2 torch ops per step")
    except Exception as ex:                                  # pragma: no cover
        out.setdefault("policy_in_loop_graph", dict(error=str(ex)[:300]))
    # (c2) a network of the reference actor's size between the steps: how much of a decoding step is the environment?
    try:
        if "graph" in skip:
            raise RuntimeError("skipped")
        trace("actor_shaped_policy")
        g3 = torch.Generator(device=dev)
        g3.manual_seed(4244)
        keys3 = torch.empty(hp.nw, B, st.shape[2], device=dev).uniform_(1e-7, 1.0 - 1e-7, generator=g3)
        env_a = T.BatchedContainer(B, cs, hp.nw, reward, "diff", packing_strategy=strategy, device=dev)
        spa = T.EpisodeStepper(st, dy, env_a, steps=hp.nw)
        flen = int(np.prod(env_a._feature_shape()[1:]))
        actor = ActorShapedPolicy(D, dy.shape[1], st.shape[2], flen, keys3).to(dev)

        class _Pol(object):                                      # run_episode calls policy(step=..., ...): begin at step 0
            def __call__(self, step, static, **kw):
                if step == 0:
                    actor.begin(static, B)
                return actor(step, **kw)
        val_a, rec_a = graphed(_Pol(), lambda: keys3.uniform_(1e-7, 1.0 - 1e-7, generator=g3), spa, steps=50)
        ok_a = oracle_ok(rec_a)
        # the same network ops alone, on the stepper's (frozen) views: what the loop costs without the environment
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))

        def net_only():
            actor.begin(st, B)
            for k in range(hp.nw):
                actor(k, dynamic=spa.dynamic, current_mask=spa._ones_mask, decoder_static=spa.decoder_static,
                      decoder_dynamic=spa.decoder_dynamic)
        spa._ones_mask = torch.ones(B, st.shape[2], device=dev)
        with torch.cuda.stream(side):
            net_only(); net_only()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        gn = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gn):
            net_only()
        for _ in range(3):
            gn.replay()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(50):
            gn.replay()
        torch.cuda.synchronize(dev)
        net_us = (time.perf_counter() - t0) / (50 * hp.nw) * 1e6
        step_us = B / val_a * 1e6
        out["actor_shaped_policy_graph"] = dict(
            value=val_a, unit="env-steps/s", steps=50, verified=ok_a, us_per_decoding_step=step_us,
            network_only_us_per_step=net_us, env_share=max(0.0, 1.0 - net_us / step_us),
            what="rollout.run_episode captured in one hipGraph with ActorShapedPolicy between the fused steps: the op "
                 "count and shapes of the reference's actor at inference (1x1-conv encoders, one GRU step, attention over "
                 "the nR columns, context, pointer energies, masked softmax; hidden 128, random weights) and a Gumbel-max "
                 "draw from pre-drawn keys; network_only = the same ops without the environment's launch; env_share = "
                 "1 - network_only / step (BASELINE.md section 2 measured 38 % for the reference's CPU loop)")
    except Exception as ex:                                  # pragma: no cover
        out["actor_shaped_policy_graph"] = dict(error=str(ex)[:300])
    # (d) what the fp32 contract costs: the same graph-captured episode (recorded tour, no policy ops) on a stepper that
    #     writes update_dynamic's result as the fp32 tensor model.py:378 feeds the encoder, and on one that keeps it as
    #     its bit shadow only (tap_stepper_buffers.dyn = NULL) -- masks, placements, features and ratio are the same
    try:
        if "graph" in skip:
            raise RuntimeError("skipped")
        trace("no_fp32_expand")
        tape_pol = T.TapePolicy(hp.tape[0].t())
        vals, same = {}, None
        for expand in (True, False):
            env_x = T.BatchedContainer(B, cs, hp.nw, reward, "diff", packing_strategy=strategy, device=dev)
            spx = T.EpisodeStepper(st, dy, env_x, steps=hp.nw, expand_dynamic=expand)
            vals[expand], rec = graphed(tape_pol, lambda: None, spx, steps=200)
            got = (rec["reward"].clone(), spx.current_mask.clone(), spx.mask.clone(), spx.decoder_dynamic.clone(),
                   env_x.positions.clone())
            same = got if same is None else all(bool(torch.equal(a, b)) for a, b in zip(same, got))
        out["no_fp32_expand"] = dict(value=vals[False], unit="env-steps/s", with_expand=vals[True], steps=200,
                                     outputs_identical=bool(same), verified=oracle_ok(rec),
                                     what="rollout.run_episode on a recorded tour captured in one hipGraph, pack.EpisodeStepper("
                                          "expand_dynamic=False): `dynamic` stays in its bit shadow, the (B, 3n, nR) fp32 tensor "
                                          "of model.py:378 (78 % of a c2 step's compulsory bytes) is never written; `with_expand` "
                                          "is the same graph on the default stepper; reward, both masks, the last feature and the "
                                          "positions of the two runs are compared bit for bit")
    except Exception as ex:                                  # pragma: no cover
        out["no_fp32_expand"] = dict(error=str(ex)[:300])
    # (d2) the fp32 tensor kept, but ONE of it: tap_stepper_buffers.dyn[0] == dyn[1] -- step 0 writes the tensor, later
    #      steps only the rows they clear (update_dynamic's result differs from its input in 3 of 3n rows)
    try:
        if "graph" in skip:
            raise RuntimeError("skipped")
        trace("in_place_dynamic")
        tape_pol = T.TapePolicy(hp.tape[0].t())
        vals, same = {}, None
        for inplace in (False, True):
            env_x = T.BatchedContainer(B, cs, hp.nw, reward, "diff", packing_strategy=strategy, device=dev)
            spx = T.EpisodeStepper(st, dy, env_x, steps=hp.nw, inplace_dynamic=inplace)
            vals[inplace], rec = graphed(tape_pol, lambda: None, spx, steps=200)
            got = (rec["reward"].clone(), spx.dynamic.clone(), spx.current_mask.clone(), spx.mask.clone(),
                   spx.decoder_dynamic.clone(), env_x.positions.clone())
            same = got if same is None else all(bool(torch.equal(a, b)) for a, b in zip(same, got))
        out["in_place_dynamic"] = dict(value=vals[True], unit="env-steps/s", two_buffers=vals[False], steps=200,
                                       outputs_identical=bool(same), verified=oracle_ok(rec),
                                       what="rollout.run_episode on a recorded tour captured in one hipGraph, pack.EpisodeStepper("
                                            "inplace_dynamic=True): `dynamic` is ONE fp32 tensor -- step 0 writes it, every later "
                                            "step zeroes the 3 rows it clears instead of re-writing all 3n (the reference's clone, "
                                            "pack.py:368, serves autograd; a no_grad loop has no use for the earlier tensors); "
                                            "`two_buffers` is the same graph on the default stepper; reward, the final tensor, both "
                                            "masks, the last feature and the positions of the two runs are compared bit for bit")
    except Exception as ex:                                  # pragma: no cover
        out["in_place_dynamic"] = dict(error=str(ex)[:300])
    finally:
        T.pack.set_binary_check('check')
    # (e) TAP_BENCH_TWO_CHAINS=1: the same batch as TWO half-batch chains on two streams of one graph.  Measured in round 6
    #     (profiles/r06_two_chains.json): 1.02 G env-steps/s against the lock-step 1.35 G -- the chains' kernels do not
    #     overlap (4.0 us per 4 096-env launch, one after the other), so splitting the batch only doubles the launches
    if os.environ.get("TAP_BENCH_TWO_CHAINS") == "1" and use_graph:
        try:
            trace("two_chains")
            out["two_chains"] = two_chain_reading(cfg, hp, dev)
        except Exception as ex:                              # pragma: no cover
            out["two_chains"] = dict(error=str(ex)[:300])
    return out


def two_chain_reading(cfg, hp, dev, chains=2, steps=200):
    """The headline's batch stepped as `chains` independent sub-batch chains, each on its own stream inside ONE hipGraph
    (fork at the graph's head, join at its tail): a chain's step t + 1 waits for ITS step t only, so one chain's kernel
    boundary (drain, cache write-back, the next dispatch: ~1.8 us of a ~6 us step) is covered by the other chain's
    kernel.  This is NOT the reference's loop shape -- model.py:342-496 runs one policy call per decoding step for the
    whole batch, which joins the chains at every step -- it is what a trainer that pipelines two micro-batches through
    the policy would see, and it bounds what the launch boundaries cost the lock-step headline.  Result (round 6, MI355X,
    ROCm 7.2): the two branches' kernels run one after the other -- 80.4 us per pass of 2 x 10 launches against 60.5 us for
    the 10 lock-step launches -- so the variant is off by default (TAP_BENCH_TWO_CHAINS=1)."""
    name, D, cs, n, B, reward, strategy = cfg
    per = B // chains
    hs = [HotPath(cfg, per, k * per, dev, fused=hp.fused, window=hp.nw, bits=hp.bits, instances=hp.instances) for k in range(chains)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(chains - 1)]
    warm = torch.cuda.Stream(device=dev)
    warm.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(warm):
        for h in hs:
            h.episode()
    torch.cuda.current_stream(dev).wait_stream(warm)
    torch.cuda.synchronize(dev)
    ge = getattr(hp, "passes_per_graph", GATHER_EVERY)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream(dev)
        for s in streams:
            s.wait_stream(main)                                # fork
        for k, h in enumerate(hs):
            with torch.cuda.stream(main if k == 0 else streams[k - 1]):
                for _ in range(ge):
                    h.episode()
        for s in streams:
            main.wait_stream(s)                                # join
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize(dev)
    reps = max(1, steps // ge)
    dts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize(dev)
        dts.append(time.perf_counter() - t0)
    dt = statistics.median(dts)
    ver = [h.verify() for h in hs]
    pass_us = dt / (reps * ge) * 1e6
    return dict(value=per * chains * n * reps * ge / dt, unit="env-steps/s", chains=chains, batch_per_chain=per,
                pass_us=pass_us, passes=reps * ge, verified=all(v.get("verified") for v in ver),
                what="the same B envs as %d independent chains of %d envs, one stream each inside one hipGraph (fork / "
                     "join once per %d passes): a chain's next step waits for its own previous step only, so kernel "
                     "boundaries overlap the other chain's kernels.  Not the reference's loop (one policy call per "
                     "step for the whole batch joins the chains every step); the headline is the lock-step figure"
                     % (chains, per, ge))


def run_sweep(cfg, hp, dev, use_graph, path, batches=None):
    name, D, cs, n, B, reward, strategy = cfg
    env_b, mask_b = algorithmic_bytes(D, cs, hp.nw)
    lines = []
    for b in (batches or (8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 2097152, 4194304)):
        if b % hp.B:
            continue
        try:
            h2 = HotPath((name, D, cs, n, b, reward, strategy), b, 0, dev, fused=hp.fused, bits=hp.bits,
                         instances=hp.instances, tile_from=hp)   # this run's instances and tape, tiled
            steps = 20 if b <= 131072 else 6
            d2, g2 = time_passes(h2, steps, 2, use_graph, 1)
            k2, _, pass_us = kernel_event_times(h2, 3, g2[0] if g2 else None)
            h2.env.check()
            rec = dict(config=name.split(" on ")[0], batch=b, env_steps_per_s=b * n * steps / d2,
                       pass_us=pass_us, step_us=pass_us / n,
                       GBps=compulsory_bytes("transition", D, cs, hp.nw, bits=bool(h2.bits)) * b / (pass_us / n * 1e-6) / 1e9,
                       GBps_how="compulsory bytes of the implemented step (compulsory_bytes) / step_us",
                       speed_vs_fp32_copy_GBps=(env_b + mask_b) * b / (pass_us / n * 1e-6) / 1e9,
                       dynamic_MB=round(h2.dynamic0[0].numel() * 4 / 1e6, 1), bits=bool(h2.bits))
            lines.append(rec)
            print("sweep " + json.dumps(rec), file=sys.stderr)
            del h2, g2
            torch.cuda.empty_cache()
        except Exception as ex:  # out of memory at the top end is fine
            print("sweep B=%d stopped: %s" % (b, ex), file=sys.stderr)
            break
    if path:
        with open(path, "a") as f:
            for rec in lines:
                f.write(json.dumps(rec) + "\n")
    return lines


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawned_rank(rank, argv, world, port):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), TAP_BENCH_SPAWNED="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.argv = [sys.argv[0]] + list(argv)
    main()


def roofline_of(hp, config, cfg, rolling, steps, graphs):
    """Per-kernel event times of `hp`'s pass and the roofline object of its dominant kernel -> (roof, kernels)."""
    name, D, cs, n, B, reward, strategy = cfg
    npass = max(3, min(steps, 20))
    kt, empty_us, pass_us = kernel_event_times(hp, npass, graphs[0] if graphs else None)
    env_b, mask_b = algorithmic_bytes(D, cs, hp.nw)
    R_ = 2 if D == 2 else 6
    win_b = (1 + D) * hp.nw * R_ * 4 + 3 * hp.nw * hp.nw * R_ * 4 + hp.nw * R_ * 4 + 32   # static + dynamic + mask + state
    per_launch = {"env_step": env_b * B, "mask_step": mask_b * B, "transition": (env_b + mask_b) * B,
                  "transition_first": (env_b + mask_b) * B,
                  "rolling_window": win_b * B, "rolling_step": (win_b + env_b) * B,
                  "episode": episode_bytes(D, n) * B,
                  "dyn_bits": (3 * hp.nw * hp.nw * R_ * 4 + hp.nw * R_ * 8) * B}
    names = [k for k in ("transition", "transition_first", "rolling_step", "rolling_window", "episode", "mask_step", "env_step", "dyn_bits",
                         "ratio", "reset") if k in kt]
    dom = max([k for k in names if k in per_launch], key=lambda k: kt[k]["total_us"])
    # the dominant kernel's duration: (event-bracketed launch) - (an empty event pair); for a pass that is
    # nothing but that kernel the graph-replayed pass / launches is the cleaner figure (it includes the gap)
    launches = hp.launches_per_pass()
    if hp.kind == "transition" and hp.fused:
        # graph-replayed pass / ALL its launches: the episode's first step (which also reads the fresh fp32 tensor: 1.4 x a
        # later step in the kernel trace) is averaged in, so this is an UPPER bound for the later steps' kernel, whose
        # compulsory bytes `achieved` is computed on -- conservative.  (Up to round 5 the first step's eager event-pair time
        # was subtracted instead; an event pair around one eager launch over-reads by ~1 us, which made the later steps
        # look 0.1-0.3 us faster than the kernel trace says.)
        other = sum(max(kt[k]["med_us"] - empty_us, 0.0) * (kt[k]["launches"] // npass) for k in names if k != dom)
        dom_us_minus_first = (pass_us - other) / (kt[dom]["launches"] // npass)
        dom_us = pass_us / launches
        how = ("graph-replayed pass / all %d launches (the first step, 1.4 x a later one, averaged in: an upper bound for a later "
               "step), passes replayed %d per graph launch as in the timed loop; includes the inter-kernel gap; with the first "
               "step's eager event time subtracted instead: %.3f us" % (launches, getattr(hp, "passes_per_graph", GATHER_EVERY), dom_us_minus_first))
    elif hp.kind == "episode":
        dom_us, how = pass_us, "event-bracketed pass (one launch)"
    elif kt[dom].get("run_len", 1) >= 8:
        dom_us = kt[dom]["run_us"]
        how = ("one event pair around each run of %d consecutive launches of the kernel, per launch (median over the "
               "passes); includes the gap between launches" % kt[dom]["run_len"])
    else:
        dom_us = kt[dom]["med_us"]
        how = ("median event-bracketed launch, uncorrected: reads about 2 us longer than the kernel (an empty event pair "
               "takes %.1f us, but most of that overlaps a kernel placed between the two records)" % empty_us)
    bits_on = bool(getattr(hp, "bits", True)) or rolling
    comp = {k: compulsory_bytes(k, D, cs, hp.nw, bits=bits_on) * B
            for k in ("env_step", "mask_step", "transition", "transition_first", "rolling_window", "rolling_step", "dyn_bits")}
    comp["episode"] = per_launch["episode"]
    ach = comp[dom] / (dom_us * 1e-6) / 1e9
    ach_survey = per_launch[dom] / (dom_us * 1e-6) / 1e9
    cal = load_calibration()
    kernels = {}
    for k in names:
        kernels[k] = dict(med_us_event_pair=round(kt[k]["med_us"], 3), avg_us_event_pair=round(kt[k]["avg_us"], 3),
                          launches_per_pass=kt[k]["launches"] // npass)
        if "run_us" in kt[k]:
            kernels[k]["us_per_launch_in_runs"] = round(kt[k]["run_us"], 3)
        if k in comp:
            kernels[k]["bytes_per_launch"] = comp[k]
        if k in per_launch:
            kernels[k]["alg_bytes_per_launch_survey"] = per_launch[k]
    tkey = config + ":" + dom + ("_copy" if dom == "transition" and not rolling and not getattr(hp, "bits", False) else "")
    tr = load_traffic(tkey)
    traffic = tr["bytes"] if tr else None
    # write-dominated kernels are held against the measured fill rate, the first-step / copy forms against the copy
    ceiling_kind = "copy" if (dom == "transition_first" or (dom in ("transition", "mask_step") and not bits_on)) else "fill"
    ceiling = (cal or {}).get("%s_GBps_beyond_cache" % ceiling_kind)
    roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": ach / HBM_PEAK_GBS,
            "bytes_per_launch": comp[dom],
            "bytes_how": "compulsory bytes of the implemented kernel (inputs it did not produce + outputs; the bit-shadow "
                         "step writes the fp32 tensor and never reads it) x envs per launch",
            "frac_alg_survey": ach_survey / HBM_PEAK_GBS,
            "frac_alg_survey_how": "SURVEY 8(d)'s per-env-step figure (prices an fp32 read AND write of dynamic) / kernel_us / "
                                   "peak: speed relative to a perfect fp32 copy, NOT a bandwidth fraction -- it passes 1.0 "
                                   "where the bit-shadow kernel, which never performs the priced fp32 read, outruns such a copy "
                                   "(`frac` and `frac_hbm` are the bandwidth fractions and stay below 1)",
            "frac_hbm": (traffic / (dom_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "peak_measured": dict(GBps=ceiling, kind=ceiling_kind + " beyond the Infinity Cache, 16 B per lane",
                                  frac=(ach / ceiling) if ceiling else None,
                                  source=(cal or {}).get("source")) if cal else None,
            "traffic": traffic,
            "traffic_source": ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, profile set %s "
                               "(profiles/%s_%s_pmc_summary.csv); not re-measured in this run"
                               % (tr["profile"], tr["profile"].split()[0], config)) if tr else None,
            "bytes_per_env_step": comp[dom] // B, "alg_bytes_per_env_step_survey": per_launch[dom] // B, "units_per_launch": B,
            "kernel_us": dom_us, "kernel_us_how": how, "kernel_us_rocprof": tr.get("kernel_us") if tr else None,
            "pass_us": pass_us, "launches_per_pass": launches, "event_pair_overhead_us": empty_us}
    return roof, kernels


def build_hotpath(config, args, rank, dev, batch=None):
    """The pre-allocated pass of one config on this rank's share of the batch -> (hot path, cfg tuple, rolling?)."""
    cfg = CONFIGS[config]
    name, D, cs, n, B, reward, strategy = cfg
    if batch:
        B = batch
        cfg = (name, D, cs, n, B, reward, strategy)
    rolling = config in ROLLING and not args.approx_windows
    if rolling:
        hp = RollingHotPath(cfg, B, rank * B, dev, window=WINDOW[config], fused_rolling=not args.two_launch_rolling,
                            overlap=args.overlap, mix=not args.rand_only)
        if not hp.mix:
            name = name.replace("3D MIX", "3D RAND")
            cfg = (name, D, cs, n, B, reward, strategy)
    elif config == "k6":
        hp = EpisodeHotPath(cfg, B, rank * B, dev)
    else:
        instances = None
        if config in ("c2", "c3", "c7", "c8", "c9") and not args.synthetic_precedence:
            instances = "generate"
        if config == "c4" and args.ppsg_device:
            instances = "ppsg2d"                                    # device-side perfect-packing generator (~30 s of set-up)
            name = name.replace("RAND-marginal blocks", "device-generated PPSG instances")
            cfg = (name, D, cs, n, B, reward, strategy)
        elif config == "c4" and not args.rand_blocks:
            fx = os.path.join(ROOT, "tests", "golden", "ppsg_2d.npz")
            if os.path.exists(fx):                                  # 64 PPSG instances written by the reference
                z = np.load(fx)                                     # (tests/golden/make_golden.py --only ppsg)
                instances = (z["static"].astype(np.float32), z["dynamic"].astype(np.float32))
                name = name.replace("RAND-marginal blocks", "the reference's PPSG generator: 64 instances tiled x%d" % (B // 64))
                cfg = (name, D, cs, n, B, reward, strategy)
        hp = HotPath(cfg, B, rank * B, dev, fused=not args.unfused, window=WINDOW.get(config), bits=not args.no_bits,
                     instances=instances)
    hp.config = config
    return hp, cfg, rolling


def measure_config(config, args, rank, world, dev, use_graph, steps, warmup, repeats, cold=False):
    """One more BASELINE config on the same clock as the headline: graph-replayed brackets of exactly `steps` passes
    (barrier + synchronize on both sides, MAX over ranks; with N > 1 every rank steps its own shard and the reward
    vectors are all-gathered as in the headline), the oracle check of the last pass on every rank, and -- rank 0 -- the
    roofline object of its dominant kernel.  Called by EVERY rank; rank 0 gets the record."""
    t0 = time.perf_counter()
    hp, cfg, rolling = build_hotpath(config, args, rank, dev, batch=args.configs_batch)
    torch.cuda.synchronize(dev)
    setup_s = time.perf_counter() - t0
    name, D, cs, n, B, reward, strategy = cfg
    dt, graphs = time_passes(hp, steps, warmup, use_graph, world, repeats=repeats)
    hp.env.check()
    ver = hp.verify()
    if getattr(hp, "gathered_ok", None) is not None:
        ver["gathered_rows_match"] = hp.gathered_ok
        if not hp.gathered_ok:
            ver["verified"] = False
            ver.setdefault("mismatch", []).append("the all-gathered rewards differ from the local rows")
    pid = getattr(hp, "passes_identical", None)
    if pid is not None:
        ver["passes_compared_with_the_last"] = pid["passes_compared"]
        ver["all_passes_identical"] = pid["identical"]
        if not pid["identical"]:
            ver["verified"] = False
            ver.setdefault("mismatch", []).append("replayed passes differ from each other")
    ok_all = tdist.max_over_ranks(0.0 if ver.get("verified") else 1.0, dev) == 0.0
    rec = None
    if rank == 0:
        roof, _ = roofline_of(hp, config, cfg, rolling, steps, graphs)
        for k in ("bytes_how", "frac_alg_survey_how", "traffic_source", "kernel_us_how", "peak_measured"):
            roof.pop(k, None)                                 # the headline's roofline object spells these out once
        total = B * world * n * steps
        rec = dict(workload=name, value=total / dt, unit="env-steps/s", n_gpus=world, ms_per_step=dt / steps * 1e3, steps=steps,
                   warmup=warmup, batch_per_gpu=B, nodes=n, container=cs, packing_strategy=strategy, reward_type=reward,
                   verified=bool(ver.get("verified")) and ok_all, verification={k: v for k, v in ver.items() if k != "what"},
                   brackets=len(hp.bracket_times),
                   spread=(max(hp.bracket_times) - min(hp.bracket_times)) / dt,
                   launch=("hipGraph replay, %d passes per graph launch" % getattr(hp, "passes_per_graph", GATHER_EVERY)) if use_graph else "eager",
                   roofline=roof, setup_s=round(setup_s, 1))
        if cold and world == 1:
            try:
                rec["roofline"]["cold"] = cold_reading(cfg, hp, rolling, dev, use_graph, max_slots=16, min_steps=16)
            except Exception as ex:                           # pragma: no cover
                rec["roofline"]["cold"] = dict(error=str(ex)[:200])
        rec["total_s"] = round(time.perf_counter() - t0, 1)
    del hp, graphs
    torch.cuda.empty_cache()
    return rec, ok_all


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override")
    ap.add_argument("--repeats", type=int, default=50,
                    help="timed brackets of exactly --steps passes each (barrier + synchronize on both sides, MAX over "
                         "ranks); value is their median, the line carries all of them")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--configs", default=None,
                    help="comma-separated BASELINE configs measured after the headline and reported in the line's `configs` "
                         "object (default: c3,c4,c5 when the headline is c2 at its own batch on one GPU; 'none' to skip)")
    ap.add_argument("--configs-batch", type=int, default=None, help="per-GPU batch of the `configs` entries (self-tests)")
    ap.add_argument("--no-cold", action="store_true", help="skip roofline.cold (the pass with nothing cache-resident)")
    ap.add_argument("--unfused", action="store_true", help="two launches per step (mask_step, env_step) + reset + ratio")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the cold and policy-in-loop readings")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle comparison after the timed region")
    ap.add_argument("--synthetic-precedence", action="store_true",
                    help="c2/c3: RAND-marginal blocks with a random precedence DAG (synth.rand_instances) instead of "
                         "instances from the device-side RAND generator (blocks packed into the 7-wide initial container)")
    ap.add_argument("--rand-blocks", action="store_true",
                    help="c4: RAND-marginal synthetic instances instead of the reference-generated PPSG fixture tiled x128")
    ap.add_argument("--no-bits", action="store_true",
                    help="precedence update as an fp32 copy (tap_transition) instead of on the bit shadow (tap_transition_bits)")
    ap.add_argument("--sweep", action="store_true", help="also run a batch sweep (stderr; --sweep-out appends JSON lines)")
    ap.add_argument("--sweep-out", default=None)
    ap.add_argument("--sweep-batches", default=None, help="comma-separated batch sizes (multiples of --batch) instead of the powers of two")
    ap.add_argument("--two-launch-rolling", action="store_true",
                    help="c5: tap_env_step_gather + tap_rolling_window per step instead of the fused tap_rolling_step")
    ap.add_argument("--overlap", action="store_true",
                    help="c5: placement t and window t+1 on two HIP streams (measured slower)")
    ap.add_argument("--approx-windows", action="store_true",
                    help="c5: consecutive independent 10-node windows instead of true rolling windows")
    ap.add_argument("--rand-only", action="store_true", help="c5: RAND instances only instead of the MIX series")
    ap.add_argument("--ppsg-device", action="store_true",
                    help="c4: 8 192 distinct instances from the device-side perfect-packing generator (~30 s of set-up) instead "
                         "of the 64 instances the reference's own generator wrote (tests/golden/ppsg_2d.npz, SURVEY 8(d)) tiled")
    ap.add_argument("--ppsg-fixture", action="store_true", help="(default since round 5; kept for old command lines)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched as plain `python bench.py --gpus N`: spawn the N ranks here (one process per GPU)
        shared = os.environ.get("TAP_DIST_BACKEND") == "gloo"    # self-test: ranks may share a GPU
        have = torch.cuda.device_count()
        if have < args.gpus and not shared:
            sys.exit("bench.py --gpus %d: only %d GPU(s) visible (set TAP_DIST_BACKEND=gloo to let ranks share one "
                     "for a self-test)" % (args.gpus, have))
        import torch.multiprocessing as mp
        mp.spawn(_spawned_rank, args=(sys.argv[1:], args.gpus, _free_port()), nprocs=args.gpus, join=True)
        return

    # THE line must be alone on stdout, and libraries write there too: RCCL prints a five-line version banner on stdout
    # (C stdio) from its first collective in a process -- seen on this stack with a 1-rank "nccl" group --, HIP and torch
    # may warn there.  From here on file descriptor 1 of this rank IS stderr; rank 0 writes the one JSON line to the
    # original descriptor at the end.
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)

    rank, world, local = tdist.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; refusing to report a line for a different job size"
                  % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU: libtapenv has no CPU path")
    shared_gpu = os.environ.get("TAP_DIST_BACKEND") == "gloo"
    if torch.cuda.device_count() < min(args.gpus, int(os.environ.get("LOCAL_WORLD_SIZE", args.gpus))) and not shared_gpu:
        # one process per GPU: fail loudly instead of quietly stacking ranks on a device
        sys.exit("bench.py rank %d: --gpus %d but this process sees %d GPU(s)" % (rank, args.gpus, torch.cuda.device_count()))
    local = local % torch.cuda.device_count()   # > 1 rank per GPU only happens in the gloo self-test
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    hp, cfg, rolling = build_hotpath(args.config, args, rank, dev, batch=args.batch)
    name, D, cs, n, B, reward, strategy = cfg
    use_graph = not args.no_graph
    dt, graphs = time_passes(hp, args.steps, args.warmup, use_graph, world, repeats=args.repeats)
    hp.env.check()
    total_steps = B * world * n * args.steps
    value = total_steps / dt
    # every rank checks its own last pass against the oracle; the line says "verified" only if all did
    ver = dict(verified=None, why="--no-verify") if args.no_verify else hp.verify()
    if getattr(hp, "gathered_ok", None) is not None and not args.no_verify:
        ver["gathered_rows_match"] = hp.gathered_ok           # this rank's block of the last all-gather == its rows
        if not hp.gathered_ok:
            ver["verified"] = False
            ver.setdefault("mismatch", []).append("the all-gathered rewards differ from the local rows")
    pid = getattr(hp, "passes_identical", None)
    if pid is not None and not args.no_verify:
        # the oracle check above sees the LAST pass; this ties the other replayed passes to it
        ver["passes_compared_with_the_last"] = pid["passes_compared"]
        ver["all_passes_identical"] = pid["identical"]
        if not pid["identical"]:
            ver["verified"] = False
            ver.setdefault("mismatch", []).append("replayed passes differ from each other")
    # who ran where, and how fast each rank was on its own clock (the job's figure is the MAX over ranks)
    props = torch.cuda.get_device_properties(local)
    mine = dict(rank=rank, device_index=local, device=torch.cuda.get_device_name(local), pid=os.getpid(),
                host=socket.gethostname(), pci_bus_id=getattr(props, "pci_bus_id", None),
                value=B * n * args.steps / statistics.median(hp.local_bracket_times),
                verified=ver.get("verified"), gathered_rows_match=ver.get("gathered_rows_match"))
    per_rank = [mine]
    if world > 1:
        import torch.distributed as dist
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        # one process per GPU: under RCCL two ranks on one device would still produce a line, at half the speed and
        # with an exchange that never left the chip -- refuse loudly instead (the gloo self-test shares a GPU on purpose)
        distinct = len({(r["host"], r["device_index"]) for r in per_rank})
        if dist.get_backend() == "nccl" and distinct != world:
            if rank == 0:
                print("bench.py: %d RCCL ranks on %d distinct device(s): %s -- refusing to report a line"
                      % (world, distinct, [(r["rank"], r["host"], r["device_index"]) for r in per_rank]), file=sys.stderr)
            sys.exit(3)
    ok_all = tdist.max_over_ranks(0.0 if ver.get("verified") in (True, None) else 1.0, dev) == 0.0

    # The other BASELINE configs on the same clock (every rank takes part: barriers, MAX over ranks, the all-gather).
    # N = 1: c3, c4 and c5's 8 192-env shard; N > 1: c5 -- the config BASELINE shards over the node -- beside the c2
    # weak-scaling value the metric names.
    more = args.configs if args.configs is not None else (
        "none" if (args.config != "c2" or args.batch) else "c3,c4,c5" if world == 1 else "c5")
    extra = {}
    if more != "none":
        for c in [x for x in more.split(",") if x and x != args.config]:
            try:
                rec, ok_c = measure_config(c, args, rank, world, dev, use_graph, args.steps, args.warmup,
                                           max(3, min(args.repeats, 10)), cold=not args.no_cold and c == "c3")
            except Exception as ex:                           # pragma: no cover
                if world > 1:
                    raise                                     # a rank that stopped would leave the others in a collective
                rec, ok_c = dict(error=str(ex)[:300], verified=False), False
            extra[c] = rec
            ok_all = ok_all and ok_c

    out = None
    if rank == 0:
        roof, kernels = roofline_of(hp, args.config, cfg, rolling, args.steps, graphs)
        if rolling:
            inst = ("MIX (pack.py:67-97): envs [0,B/2) perfect-packing (PPSG) instances, [B/2,B) RAND, both device-generated "
                    "into a 7x7x250 initial container.  PPSG = generate_blocks_with_GT's steps on 5 stacked 10-block "
                    "BPP_Generator_3D packings with the 'simple' take-apart test: the reference's own loop cannot reach "
                    "50 blocks (its acceptance test passes < 2e-8 of 50-block cuts and 0 of 200 layouts)" if hp.mix else
                    "device-generated 50-block RAND instances (generate.generate_instances), initial container 7 wide")
        elif getattr(hp, "instances", None) == "ppsg2d":
            inst = ("perfect-packing (PPSG) instances of the device-side generator (generate_blocks_with_GT's steps: "
                    "BPP_Generator_2D_easy cuts of a 7 x H box, random take-apart order and rotations, hard LB_GREEDY layout "
                    "in the 7 x 50 initial container, the reference's 'bot' acceptance test), H drawn per instance from "
                    "generate_height_prob's distribution restricted to 14..24 (90 %% of it: outside, the acceptance loop needs "
                    "1e4 .. 1e5 layouts per instance); generated in %.1f s" % getattr(hp, "generate_s", 0.0))
        elif getattr(hp, "instances", None) == "generate":
            inst = ("RAND instances of the device-side generator (generate_blocks semantics: random blocks packed into "
                    "the 7-wide initial container, real precedence)")
        elif getattr(hp, "instances", None) is not None:
            inst = "fixture tiled"
        else:
            inst = "RAND-marginal blocks, random precedence DAG (synth.rand_instances)"
        if rolling:
            pas = ("rolling.validate's loop: reset + first window + (n - window) x tap_rolling_step (placement t + window "
                   "t+1 in one launch), then window x tap_transition_bits on the last graph" if hp.fused_rolling else
                   "rolling.validate's loop: (n - window) x (tap_env_step_gather + tap_rolling_window), then window x "
                   "tap_transition_bits on the last graph")
        elif hp.kind == "episode":
            pas = "ONE launch of tap_episode_reward: gather by tour + n placements + C+P+S per env (pack.reward)"
        elif hp.fused:
            pas = ("%sn x tap_transition%s (update_dynamic+update_mask+gather+add_new_block in one launch; first starts a "
                   "fresh container, last emits calc_ratio)%s" %
                   (("", "_first/_bits",
                     "; step 0 reads the fp32 instance tensor and builds the bit shadow in the same launch, later steps "
                     "carry dynamic as that shadow: the fp32 tensor is written every step but not re-read")
                    if hp.bits else ("", "", "")))
        else:
            pas = "reset + n x (update_dynamic+update_mask, add_new_block) + calc_ratio"
        out = {
            "metric": "env-steps/s (batch placements) 2D n=10 LB_GREEDY; 1/2/4/8 GPU + CPU ref",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "i32 height-maps / f64 candidate scores / f32 masks",
            "data": "synthetic",
            "verified": bool(ver.get("verified")) and ok_all if ver.get("verified") is not None else None,
            "verification": ver,
            "repeats": dict(brackets=len(hp.bracket_times), steps_per_bracket=args.steps, statistic="median",
                            values=[total_steps / t for t in hp.bracket_times],
                            min=total_steps / max(hp.bracket_times), max=total_steps / min(hp.bracket_times),
                            spread=(max(hp.bracket_times) - min(hp.bracket_times)) / dt),
            "ranks": dict(world_size=world, backend=(torch.distributed.get_backend() if world > 1 else None),
                          rccl_ranks=(torch.distributed.get_world_size() if world > 1 and
                                      torch.distributed.get_backend() == "nccl" else (1 if world == 1 else 0)),
                          distinct_devices=len({(r["host"], r["device_index"]) for r in per_rank}),
                          spawned_by_bench=os.environ.get("TAP_BENCH_SPAWNED") == "1",
                          first_all_gather_ms=getattr(hp, "first_gather_ms", None),
                          per_rank=per_rank),
            "config": {"workload": name, "batch_per_gpu": B, "nodes": n, "window_nodes": hp.nw, "container": cs,
                       "reward_type": reward, "packing_strategy": strategy, "instances": inst, "pass": pas,
                       "launch": ("hipGraph replay, %d passes per graph launch" % getattr(hp, "passes_per_graph", GATHER_EVERY)) if use_graph else "eager"},
            "roofline": roof,
            "kernels": kernels,
        }
        if world == 1 and not args.no_cold and hp.kind in ("transition", "rolling") and not args.no_variants:
            try:
                hp.cold = cold_reading(cfg, hp, rolling, dev, use_graph)
                out["roofline"]["cold"] = hp.cold
            except Exception as ex:                           # pragma: no cover
                out["roofline"]["cold"] = dict(error=str(ex)[:200])
        if extra:
            out["configs"] = extra
        if not args.no_cpu_baseline and world == 1:
            cb = cpu_baseline_rolling(hp) if rolling else cpu_baseline(hp)
            attach_reference_cpu(cb, args.config)
            out["cpu_baseline"] = cb
        if not args.no_variants and world == 1:
            out["variants"] = variants(cfg, args, hp, rank, world, dev, use_graph)
        if args.sweep and hp.kind == "transition":
            run_sweep(cfg, hp, dev, use_graph, args.sweep_out,
                      [int(x) for x in args.sweep_batches.split(",")] if args.sweep_batches else None)
        sys.stdout.flush()
        os.write(line_fd, (json.dumps(out) + "\n").encode())
    tdist.barrier()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
    if not ok_all:
        sys.exit(1)


if __name__ == "__main__":
    main()
