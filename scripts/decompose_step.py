#!/usr/bin/env python3
"""Where the time of one fused decoding step (k_transition, the bit-shadow form) goes -- VERDICT r3 item 2.

(1) DECOMPOSITION (-DTAP_PROF_SWITCH build: the product kernel plus two flag bits that make one kind of wave return
    at once): per-launch time, inside a hipGraph of whole passes as bench.py times them, of
      empty      every wave returns at entry ("empty kernel of the geometry": grid, workgroup size, LDS, kernarg)
      place      placement waves only
      mask       stream waves only (update_dynamic + update_mask on the bit shadow)
      fused      the product step
      fused_inplace   the same with ONE fp32 output buffer for every step instead of two alternating ones
(2) TIMELINE (-DTAP_PROF build: clock stamps, see tap_masks.h): when, relative to the first wave's entry, the waves
    of a launch start, have their inputs, issue their last store and see it acknowledged.

Build recipe (the product build has neither):
    cd tap-net_amd/csrc && mkdir -p ../../build_prof && for v in PROF PROF_SWITCH; do \\
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -I../../include -I. -DTAP_$v \\
            -c transition.hip -o ../../build_prof/transition_$v.o && \\
      hipcc --offload-arch=gfx950 -shared $(ls build/*.o | grep -v /transition.o) ../../build_prof/transition_$v.o \\
            -o ../../build_prof/libtapenv_$v.so; done
Run (GPU box):  python scripts/decompose_step.py --out profiles/r04_step_decomposition.json
(the script re-executes itself once per library: TAP_LIB_PATH is read at import).
"""
import argparse, ctypes as C, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NOSTREAM, NOPLACE = 1 << 8, 1 << 9
SHAPES = {"c2": (2, [5, 50], 10, 8192), "c3": (3, [5, 5, 50], 10, 4096)}


def setup(shape, seed=1):
    import numpy as np, torch
    import tap_net_amd as T
    from tap_net_amd import _lib, synth
    D, cs, n, B = SHAPES[shape]
    dev = torch.device("cuda:0")
    static, dynamic = synth.rand_instances(B, n, D, seed=seed)
    tape = synth.random_feasible_tape(static, dynamic, n, seed=seed + 1).t().contiguous().to(dev)    # (n, B)
    st, dy = static.to(dev), dynamic.to(dev)
    env = T.BatchedContainer(B, cs, n, "C+P+S-lb-soft", "diff", device=dev)
    R = 2 if D == 2 else 6
    nR, rows = n * R, 3 * n
    f32 = dict(dtype=torch.float32, device=dev)
    buf = dict(bits=[torch.empty(B, nR, dtype=torch.int64, device=dev) for _ in range(2)],
               dyn=[torch.empty(B, rows, nR, **f32) for _ in range(2)],
               cur=[torch.empty(B, nR, **f32) for _ in range(2)], mask=[torch.ones(B, nR, **f32) for _ in range(2)],
               ones=torch.ones(B, nR, **f32), feat=env._new_feature(), ratio=torch.empty(B, **f32), cnt=torch.zeros(1, dtype=torch.int32, device=dev))
    L, ctx = _lib.lib(), _lib.ctx(dev)
    P = lambda t: C.c_void_p(t.data_ptr() if t is not None else None)   # noqa: E731

    def step(t, extra=0, inplace=False, stream=None, nodyn=False):
        s = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        w, r = t & 1, (t & 1) ^ 1
        out = None if nodyn else buf["dyn"][0] if inplace else buf["dyn"][w]
        flags = (1 if t == 0 else 0) | (2 if t == n - 1 else 0) | extra
        if t == 0:
            rc = L.tap_transition_first(ctx, C.byref(env.desc), P(env._state), n, R, rows, 3, P(dy), P(st), st.shape[1], P(tape[0]),
                                        P(buf["ones"]), P(buf["bits"][w]), P(out), P(buf["cur"][w]), P(buf["mask"][w]),
                                        P(buf["feat"]), P(buf["ratio"]), P(buf["cnt"]), flags, s)
        else:
            rc = L.tap_transition_bits(ctx, C.byref(env.desc), P(env._state), n, R, rows, 3, P(buf["bits"][r]), P(st), st.shape[1],
                                       P(tape[t]), P(buf["mask"][r]), P(buf["bits"][w]), P(out), P(buf["cur"][w]), P(buf["mask"][w]),
                                       P(buf["feat"]), P(buf["ratio"]), flags, s)
        assert rc == 0, rc
    return dict(torch=torch, np=np, L=L, n=n, B=B, dev=dev, step=step, buf=buf, env=env, D=D, cs=cs)


def run_switch(shape, slots=1):
    """slots > 1: the COLD form -- pass i of the graph runs on instance set i % slots (own instances, tape, containers and
    every output buffer), the sets together several times the 256 MB Infinity Cache (bench.py: roofline.cold)."""
    sets = [setup(shape, seed=1 + 10 * k) for k in range(slots)]
    S = sets[0]
    torch, n = S["torch"], S["n"]
    res = {}
    PASSES = 16 if slots == 1 else slots
    for name, extra, inplace, nodyn in (("empty", NOSTREAM | NOPLACE, False, False), ("place", NOSTREAM, False, False),
                                        ("mask", NOPLACE, False, False), ("mask_no_fp32", NOPLACE, False, True),
                                        ("fused", 0, False, False), ("fused_no_fp32", 0, False, True),
                                        ("fused_inplace", 0, True, False)):
        def one_pass(k=0):
            for t in range(n):
                sets[k % slots]["step"](t, extra, inplace, nodyn=nodyn)
        side = torch.cuda.Stream(device=S["dev"])
        side.wait_stream(torch.cuda.current_stream(S["dev"]))
        with torch.cuda.stream(side):
            for k in range(slots):
                one_pass(k); one_pass(k)
        torch.cuda.current_stream(S["dev"]).wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for k in range(PASSES):
                one_pass(k)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        vals = []
        for _ in range(7):
            t0 = time.perf_counter()
            for _ in range(10):
                g.replay()
            torch.cuda.synchronize()
            vals.append((time.perf_counter() - t0) / (10 * PASSES * n) * 1e6)
        vals.sort()
        res[name] = dict(us_per_launch=round(vals[len(vals) // 2], 3), min=round(vals[0], 3), max=round(vals[-1], 3))
    res["how"] = ("per-launch time of a graph of %d passes x %d launches (step 0 = tap_transition_first, the others "
                  "tap_transition_bits) over %d instance set(s), median of 7 brackets of 10 replays; the first step's fp32 read is "
                  "in the average; *_no_fp32 = dyn_out NULL (the step keeps `dynamic` as its bit shadow only)" % (PASSES, n, slots))
    return res


def run_timeline(shape):
    S = setup(shape)
    torch, np, n, B = S["torch"], S["np"], S["n"], S["B"]
    WGS, WAVES = 2048, 16                                       # tap_masks.h: TAP_PROF_WGS, TAP_PROF_WAVES
    raw = (C.c_ulonglong * (WGS * WAVES * 4))()
    rd = S["L"].tap_prof_read_timeline
    for t in range(n):                                          # warm-up pass
        S["step"](t)
    torch.cuda.synchronize()
    out = {}
    for rep in range(3):
        for t in range(n):
            rd(raw, 1)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); S["step"](t); e1.record()
            torch.cuda.synchronize()
            if rep == 2 and t in (1, 5, 9):
                rd(raw, 0)
                a = np.frombuffer(raw, dtype=np.uint64).reshape(WGS, WAVES, 4).astype(np.int64)
                used = a[:, :, 0] > 0
                # s_memrealtime is per XCD: the eight counters are offset against each other by microseconds (the offsets
                # differ from box to box), so every XCD's waves (block b runs on XCD b % 8) are timed against that XCD's
                # own first wave
                rel = np.zeros(a.shape, dtype=np.float64)
                for x in range(8):
                    ux = used[x::8]
                    if ux.any():
                        rel[x::8] = (a[x::8] - a[x::8, :, 0][ux].min()) * 10.0 / 1000.0    # 100 MHz ticks -> us
                nwg = int(used.any(1).sum())
                env_waves = 1 if S["D"] == 2 else 4             # TransGeom: 8 envs x G lanes / 64 (G = 8 at W = 5, 32 at 5 x 5)
                place = used & (np.arange(WAVES)[None, :] < env_waves)
                stream = used & ~place
                def q(m, i):
                    v = rel[:, :, i][m]
                    return dict(min=round(float(v.min()), 2), p50=round(float(np.median(v)), 2), p90=round(float(np.percentile(v, 90)), 2),
                                max=round(float(v.max()), 2))
                ent = rel[:, :, 0]
                wg_entry = np.where(used, ent, np.nan)
                by_xcd = [round(float(np.nanmax(wg_entry[x::8][:nwg // 8 + 1])), 2) for x in range(8)]
                dec = [round(float(np.nanmedian(wg_entry[i * nwg // 10:(i + 1) * nwg // 10])), 2) for i in range(10)]
                hist = np.histogram(ent[used], bins=np.arange(0, 3.25, 0.25))[0].tolist()
                out["step%d" % t] = dict(workgroups=nwg, last_entry_by_xcd=by_xcd, entry_median_by_blockidx_decile=dec,
                                         entry_histogram_quarter_us=hist, event_pair_us=round(e0.elapsed_time(e1) * 1e3, 2),
                                         stream=dict(entry=q(stream, 0), inputs_arrived_first_store=q(stream, 1),
                                                     last_store_issued=q(stream, 2), stores_acknowledged=q(stream, 3)),
                                         placement=dict(entry=q(place, 0), state_and_block_loaded=q(place, 1),
                                                        placement_decided=q(place, 2), results_stored=q(place, 3)))
    out["how"] = ("us since the first wave's entry ON THE SAME XCD (s_memrealtime, 100 MHz: 0.01 us steps; the counters of the eight "
                  "XCDs are not aligned with each other), over all waves of one eager launch; "
                  "the stamps add an s_waitcnt vmcnt(0) before the first store, so the launch is ~0.3 us slower than the product's")
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--child", default=None)
    ap.add_argument("--shape", default="c2")
    ap.add_argument("--slots", type=int, default=1)
    ap.add_argument("--no-timeline", action="store_true")
    a = ap.parse_args()
    if a.child:
        r = run_switch(a.shape, a.slots) if a.child == "switch" else run_timeline(a.shape)
        print("RESULT " + json.dumps(r))
        sys.exit(0)
    allr = {}
    for shape in ("c2", "c3"):
        allr[shape] = {}
        kinds = [("switch", "libtapenv_PROF_SWITCH.so", 1), ("switch_cold", "libtapenv_PROF_SWITCH.so", 20 if shape == "c2" else 14)]
        if not a.no_timeline:
            kinds.append(("timeline", "libtapenv_PROF.so", 1))
        for kind, lib, slots in kinds:
            env = dict(os.environ, TAP_LIB_PATH=os.path.join(ROOT, "build_prof", lib))
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", kind.split("_")[0], "--shape", shape,
                                "--slots", str(slots)], env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(p.stdout[-2000:], p.stderr[-3000:], file=sys.stderr)
                allr[shape][kind] = dict(error=p.stderr[-500:])
                continue
            allr[shape][kind] = json.loads(line[0][7:])
    txt = json.dumps(allr, indent=1)
    print(txt)
    if a.out:
        open(a.out, "w").write(txt + "\n")
