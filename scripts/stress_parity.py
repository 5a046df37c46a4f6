"""Randomised parity sweep on the GPU: random container shapes / block counts / reward strings for the
placement families (LB_GREEDY 2D/3D, MACS 2D/3D, legacy LB 2D/3D), every env compared with the CPU oracle
(positions, stable flags, final height-map, fp64 ratio) and the number of flagged containers compared
with the number of envs in which the reference would raise.

    python scripts/stress_parity.py 3000 [out.json]   # round 1: 93.6 M env-steps, 0 mismatching envs

Prints one CASE line per configuration that disagrees and a final JSON summary (also written to out.json):
configurations, env-steps, mismatching envs, envs the oracle flags (the reference raises there) and whether
the library flagged the same number.
"""
import os, sys, time, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import tap_net_amd as T
DEV = "cuda:0"
def one(cs, n, reward, strategy, B, lo, hi, seed, feat="diff"):
    D = len(cs)
    rng = np.random.RandomState(seed)
    blocks = rng.randint(lo, hi, size=(B, n, D)).astype(np.int32)
    ref = O.run_episodes(O.make_desc(cs, n, reward, feat, strategy), blocks, nthreads=8)
    good = ref["errs"] == 0
    env = T.BatchedContainer(B, cs, n, reward, feat, packing_strategy=strategy, device=DEV)
    blk = torch.as_tensor(blocks, device=DEV)
    for t in range(n):
        env.add_new_blocks(blk[:, t].contiguous())
    pos = env.positions.cpu().numpy(); st = env.stable.cpu().numpy().astype(np.uint8)
    hm = env.heightmap.cpu().numpy().reshape(B, -1); r = env.calc_ratios64().cpu().numpy()
    bad = ~((pos == ref["positions"]).all((1, 2)) & (st == ref["stable"]).all(1) & (hm == ref["heightmaps"][:, n - 1]).all(1)
            & ((r == ref["ratio"]) | (np.isnan(r) & np.isnan(ref["ratio"]))))
    bad &= good
    try:
        env.check(); flagged = 0
    except IndexError as e:
        import re
        m = re.search(r"(\d+) container", str(e)); flagged = int(m.group(1)) if m else -1
    except Exception as e:
        flagged = -2
    return int(bad.sum()), int((~good).sum()), flagged
t0 = time.time(); total = 0; nbad = 0; nflag_mismatch = 0; noracle_err = 0
fam = {}
cases = []
only = os.environ.get("STRESS_ONLY_MOD20")          # e.g. "2,3": only the seeds with these residues mod 20
only25 = os.environ.get("STRESS_ONLY_MOD25")        # e.g. "1,6": only the seeds with these residues mod 25 (the wide MACS 2D cases)
for seed in range(int(sys.argv[1])):
    if only and str(seed % 20) not in only.split(","): continue
    if only25 and str(seed % 25) not in only25.split(","): continue
    rs = np.random.RandomState(1000 + seed)
    kind = seed % 5
    if kind == 4:    # legacy LB, 2D and 3D (voxel-level kernel)
        if rs.rand() < 0.5:
            W = int(rs.randint(1, 12)); cs = [W, int(rs.choice([40, 60, 120]))]
        else:
            cs = [int(rs.randint(1, 9)), int(rs.randint(1, 9)), int(rs.choice([40, 60, 120]))]
        n = int(rs.randint(4, 20)); hi = int(rs.randint(2, 7))
        reward = str(rs.choice(["C+P+S-lb-soft", "C+P+S-lb-hard", "C+P-lb-soft", "C+P-lb-hard"])); strat = "LB"
    elif kind == 0:    # MACS 3D
        W, L = rs.randint(2, 8), rs.randint(2, 8)
        cs = [int(W), int(L), int(rs.choice([40, 64, 100, 200]))]; n = int(rs.randint(6, 22)); hi = int(min(W, L, 5)) + 1
        reward = str(rs.choice(["C+P+S-mcs-soft", "C+P+S-mcs-hard", "C+P+S-mul-soft", "C+P+S-mul-hard", "mcs-soft", "mcs-hard", "C+P-mcs-soft"])); strat = "MACS"
        if seed % 20 == 0:                                   # one in four of the family: 30 .. 70 small blocks into a tall container --
            r2 = np.random.RandomState(7000 + seed)          # the voxel-identity masks of tap_macs3_place change regime at 32 and 64
            n = int(r2.randint(30, 71)); cs[2] = int(r2.choice([300, 600, 1000])); hi = 3   # placed blocks (narrow -> by_bits -> fallback)
        if seed % 20 == 10:                                  # and one in four above 64 cells / a side above 8 (macs3_big.hip:
            r2 = np.random.RandomState(8000 + seed)          # one thread per container, rows as 64-bit masks)
            W, L = int(r2.randint(2, 24)), int(r2.randint(9, 24))
            if r2.rand() < 0.5: W, L = L, W
            cs = [W, L, int(r2.choice([30, 60, 100]))]; n = int(r2.randint(8, 40)); hi = int(min(W, L, r2.randint(3, 17))) + 1   # round 5: sides up to 16 (tap_stable_wide.h)
    elif kind == 1:  # MACS 2D
        W = int(rs.randint(2, 14)); cs = [W, int(rs.choice([60, 100, 200]))]; n = int(rs.randint(6, 24)); hi = min(W, 6) + 1
        reward = str(rs.choice(["C+P+S-mcs-soft", "C+P+S-mcs-hard", "C+P+S-mul-soft", "mcs-soft", "C+P-mcs-hard"])); strat = "MACS"
        if seed % 25 == 1:                                   # one in five of the family: 17 .. 64 columns (tap_macs_wide.h)
            W = int(np.random.RandomState(5000 + seed).randint(17, 65)); cs[0] = W; hi = int(np.random.RandomState(6000 + seed).randint(4, 12))
        if seed % 25 == 6:                                   # and one in five above 64 columns (macs_big.hip)
            W = int(np.random.RandomState(5000 + seed).randint(65, 160)); cs[0] = W; hi = int(np.random.RandomState(6000 + seed).randint(4, 16))
    elif kind == 2:  # LB 3D
        W, L = rs.randint(1, 9), rs.randint(1, 9)
        cs = [int(W), int(L), int(rs.choice([60, 120, 250]))]; n = int(rs.randint(4, 30)); hi = int(rs.randint(2, 8))
        reward = str(rs.choice(["C+P+S-lb-soft", "C+P+S-lb-hard", "C+P-lb-soft", "C+P-lb-hard"])); strat = "LB_GREEDY"
        if seed % 20 == 2:                                   # one in four: above 64 cells / a side above 8 (big.hip: one wavefront
            r2 = np.random.RandomState(9000 + seed)          # per container, soft and hard)
            W, L = int(r2.randint(2, 26)), int(r2.randint(9, 26))
            if r2.rand() < 0.5: W, L = L, W
            cs = [W, L, int(r2.choice([40, 80, 160]))]; n = int(r2.randint(6, 40)); hi = int(min(W, L, r2.randint(3, 17))) + 1   # round 5: sides up to 16
    else:            # LB 2D
        W = int(rs.randint(1, 40)); cs = [W, int(rs.choice([60, 120, 250]))]; n = int(rs.randint(4, 30)); hi = int(rs.randint(2, 10))
        reward = str(rs.choice(["C+P+S-lb-soft", "C+P+S-lb-hard", "C+P-lb-soft", "C+P-lb-hard"])); strat = "LB_GREEDY"
        if seed % 20 == 3:                                   # one in four: 65 .. 400 columns (big.hip)
            r2 = np.random.RandomState(9500 + seed)
            W = int(r2.randint(65, 401)); cs[0] = W; hi = int(r2.randint(3, 40)); n = int(r2.randint(6, 40))
    B = 2048
    feat = str(rs.choice(["diff", "zero", "full"]))
    b, ne, fl = one(cs, n, reward, strat, B, 1, hi, seed, feat)
    total += B * n; nbad += b; noracle_err += ne
    f = fam.setdefault(("MACS" if strat == "MACS" else "LB (legacy)" if strat == "LB" else "LB_GREEDY") + (" 3D" if len(cs) == 3 else " 2D"), dict(configurations=0, env_steps=0, mismatching_envs=0))
    f["configurations"] += 1; f["env_steps"] += B * n; f["mismatching_envs"] += b
    if b or (fl != ne and not (ne == 0 and fl == 0)):
        nflag_mismatch += int(fl != ne)
        print("CASE", cs, n, reward, strat, "hi", hi, "mismatch", b, "oracle-err", ne, "flagged", fl)
import json
summary = dict(script="scripts/stress_parity.py", configurations=sum(f["configurations"] for f in fam.values()), seeds=int(sys.argv[1]), only_seeds_mod_20=only, only_seeds_mod_25=only25, envs_per_configuration=2048, env_steps=total,
               mismatching_envs=nbad, envs_where_the_reference_raises=noracle_err,
               configurations_with_a_different_flagged_count=nflag_mismatch, families=fam, seconds=round(time.time() - t0, 1),
               compared="positions, stable flags, final height-map, fp64 calc_ratio (bit pattern) per env; flagged-container count")
print(json.dumps(summary))
if len(sys.argv) > 2:
    json.dump(summary, open(sys.argv[2], "w"), indent=1)
