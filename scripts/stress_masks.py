"""Randomised parity sweep of the DECODING STEP's precedence half on the GPU (pack.update_dynamic / pack.update_mask,
pack.py:276-376, as the fused step's stream waves compute them): random window shapes -- D, n, 'bot' / 'rot' input types, batches
that are and are not multiples of a workgroup's envs -- random 0/1 precedence tensors and random tapes, through the step object
in its three forms (two alternating fp32 tensors, ONE tensor updated in place, no fp32 tensor) and checked after EVERY step
against the CPU oracle: the fp32 tensor, its bit shadow, both masks; at the end of the episode the reward.

    python scripts/stress_masks.py 300 [out.json]

Complements scripts/stress_parity.py (placements) and scripts/stress_rolling.py (rolling windows)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import tap_net_amd as T
from tap_net_amd import pack, synth
DEV = "cuda:0"


def bits_of(dyn):
    """(B, rows, nR) 0/1 -> (B, planes, nR) int64 words, bit r of plane r // 64"""
    B, rows, nR = dyn.shape
    planes = (rows + 63) // 64
    out = np.zeros((B, planes, nR), np.uint64)
    for r in range(rows):
        out[:, r // 64] |= (dyn[:, r] != 0).astype(np.uint64) << np.uint64(r % 64)
    return out.view(np.int64)


def one(seed):
    rs = np.random.RandomState(5000 + seed)
    D = int(rs.choice([2, 3]))
    R = 2 if D == 2 else 6
    kind = seed % 4
    if kind == 0:
        n = 10                                                   # the compiled-in windows
    else:
        n = int(rs.choice([2, 4, 6, 8, 12, 14, 16, 18, 20])) if D == 2 else int(rs.choice([2, 4, 6, 8, 10, 12, 14]))
    input_type = "rot" if (kind == 3 and rs.rand() < 0.5) else "bot"
    B = int(rs.choice([8, 64, 1024, 4096])) if rs.rand() < 0.4 else int(rs.randint(1, 3000))
    # the container family decides WHICH fused kernel carries the stream waves: lane-per-cell LB_GREEDY / MACS (k_transition,
    # k_transition_macs*), one wavefront per container (k_big_transition, k_macs2d_wave_transition, k_macs3d_wave_transition)
    fam = int(rs.randint(0, 6)) if seed % 2 else 0
    H = 4 * n + 8
    strategy, reward = "LB_GREEDY", "C+P+S-lb-soft"
    if fam == 0:
        cs = [5, H] if D == 2 else [5, 5, H]
    elif fam == 1:
        cs = [int(rs.randint(6, 17)), H] if D == 2 else [int(rs.randint(4, 9)), int(rs.randint(4, 9)), H]
        cs = cs if D == 2 or cs[0] * cs[1] <= 64 else [8, 8, H]
    elif fam == 2:                                               # LB_GREEDY above 64 cells
        cs = [int(rs.choice([70, 100])), H] if D == 2 else [int(rs.choice([9, 10, 12])), int(rs.choice([9, 10])), H]
    elif fam == 3:                                               # MACS on lanes
        strategy, reward = "MACS", "C+P+S-mcs-soft"
        cs = [int(rs.randint(5, 17)), H] if D == 2 else [5, 5, H]
    elif fam == 4:                                               # MACS, one wavefront per container
        strategy, reward = "MACS", str(rs.choice(["C+P+S-mcs-soft", "mcs-soft"]))
        cs = [int(rs.choice([20, 40, 64])), H] if D == 2 else [int(rs.choice([9, 10])), int(rs.choice([9, 10])), H]
    else:
        strategy, reward = "MACS", "C+P+S-mul-hard"
        cs = [7, H] if D == 2 else [6, 6, H]
    if fam in (2, 4):
        B = min(B, 600)                                          # the oracle's MACS / big-container episodes are the slow part
    static, dynamic = synth.rand_instances(B, n, D, seed=seed)
    tape = synth.random_feasible_tape(static, dynamic, n, seed=seed + 1)
    if input_type == "rot":
        dynamic = dynamic[:, :n].contiguous()
    st, dy, tp = static.to(DEV), dynamic.to(DEV), tape.to(DEV)
    stn, tpn = static.numpy(), tape.numpy()
    ur = 3 if input_type == "bot" else 1
    bad = []
    modes = [dict(), dict(inplace_dynamic=True), dict(expand_dynamic=False)]
    for mi, kw in enumerate(modes):
        env = T.BatchedContainer(B, cs, n, reward, "diff", packing_strategy=strategy, device=DEV)
        try:
            sp = pack.EpisodeStepper(st, dy, env, input_type=input_type, **kw)
        except ValueError:
            continue                                             # no bit shadow for this shape: not this sweep's subject
        for with_mask in ((True, False) if seed % 3 == 0 else (bool(seed & 1),)):
            sp.begin(st, dy, initial_mask=with_mask)
            dyn = dynamic.numpy().copy()
            if input_type == "bot":
                cur = O.initial_mask(dyn, n)
            else:
                cur = O.initial_mask(np.concatenate([dyn, np.zeros_like(dyn), np.zeros_like(dyn)], 1), n)
            mask = np.ones_like(cur)
            if with_mask and not np.array_equal(sp.current_mask.cpu().numpy(), cur):
                bad.append((mi, with_mask, -1, "initial mask"))
            for t in range(n):
                sp.step(tp[:, t].contiguous())
                dyn = O.update_dynamic(dyn, stn, tpn[:, t], n, ur)
                cur, mask = O.update_mask(mask, dyn if input_type == "bot" else np.concatenate([dyn, np.zeros_like(dyn), np.zeros_like(dyn)], 1), tpn[:, t], n, R)
                if sp.dynamic is not None and not np.array_equal(sp.dynamic.cpu().numpy(), dyn):
                    bad.append((mi, with_mask, t, "dynamic"))
                if not np.array_equal(sp.dynamic_bits.cpu().numpy().reshape(B, -1, n * R), bits_of(dyn)):
                    bad.append((mi, with_mask, t, "bits"))
                if not np.array_equal(sp.current_mask.cpu().numpy(), cur) or not np.array_equal(sp.mask.cpu().numpy(), mask):
                    bad.append((mi, with_mask, t, "masks"))
            blocks = np.stack([stn[np.arange(B), 1:, tpn[:, t]] for t in range(n)], axis=1).astype(np.int32)
            ref = O.run_episodes(O.make_desc(cs, n, reward, "diff", strategy), blocks, nthreads=8, want_features=False, want_heightmaps=False)
            ok = ref["errs"] == 0
            if not np.array_equal(sp.ratio.cpu().numpy()[ok], ref["ratio"].astype(np.float32)[ok]):
                bad.append((mi, with_mask, n, "ratio"))
        try:
            sp.check()
        except IndexError:
            pass                                                 # containers the reference raises in too (oracle: errs != 0)
    return dict(D=D, n=n, B=B, input_type=input_type, cs=cs, strategy=strategy, steps=B * n * len(modes)), bad


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    t0 = time.time()
    pack.set_binary_check('trust')
    total, nbad, fam = 0, 0, {}
    for seed in range(N):
        meta, bad = one(seed)
        total += meta["steps"]
        key = "%dD %s %s n%s%s" % (meta["D"], meta["strategy"], meta["input_type"], "=10" if meta["n"] == 10 else "!=10",
                                     " big" if (meta["cs"][0] > 16 or (meta["D"] == 3 and meta["cs"][0] * meta["cs"][1] > 64)) else "")
        fam[key] = fam.get(key, 0) + 1
        if bad:
            nbad += 1
            print("CASE", seed, meta, bad[:6], flush=True)
    out = dict(configurations=N, env_steps=total, mismatching_configurations=nbad, families=fam, seconds=round(time.time() - t0, 1),
               what="step object in three forms (two fp32 tensors / one in place / none) vs the oracle after every step: dynamic, bit shadow, "
                    "both masks, final ratio")
    print(json.dumps(out))
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], "w"), indent=1)
    sys.exit(1 if nbad else 0)


if __name__ == "__main__":
    main()
