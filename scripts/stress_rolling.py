"""Randomised parity sweep of the rolling-window path on the GPU: random (D, N, window, container width) --
windows of 2 .. 24 nodes over 4 .. 64 blocks (wave-per-instance kernels: closed-form, half-closed and emulated
CPython-set orders; thread-level set order above 18 nodes) and up to 330 blocks (NW-word graphs on one wavefront up to
256 blocks with windows of at most 64 / NW nodes, the thread-per-instance path otherwise) --
every window tensor, node list, initial mask and the final packing of a slice of the instances against the
CPU oracle (generate.InitialContainer + tools.Container restatements), fused and two-launch forms.

    python scripts/stress_rolling.py 300 [out.json]
"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import tap_net_amd as T
from tap_net_amd import generate as gen
DEV = "cuda:0"


def one(D, N, child, W, B, seed, fused):
    rs = np.random.RandomState(seed)
    init = [W, min(6 * N + 10, 4000)] if D == 2 else [W, W, min(6 * N + 10, 4000)]
    hi = min(5, W + 1)
    blocks = torch.as_tensor(rs.randint(1, hi, size=(B, N, D)).astype(np.int32), device=DEV)
    positions, _, _ = gen.pack_blocks(blocks, init, 'C+P+S-lb-soft')
    g = torch.Generator(device=DEV); g.manual_seed(seed)
    seen = {}

    def policy(step, static, dynamic, current_mask, **_):
        seen[step] = (static.cpu().numpy(), dynamic.cpu().numpy(), current_mask.cpu().numpy())
        return torch.multinomial(current_mask, 1, generator=g).squeeze(1)
    H = min(4 * N + 10, 4000)
    out = T.run_rolling_episode(blocks, positions, init, policy, 5, H, child_graph_size=child, fused=fused)
    out["env"].check(); out["windows"].check()
    tour = out["tour_idx"].cpu().numpy(); picked = out["nodes"].cpu().numpy()
    bl, ps = blocks.cpu().numpy(), positions.cpu().numpy()
    cs = [5, H] if D == 2 else [5, 5, H]
    bad = 0
    for b in range(0, B, max(1, B // 6)):
        ro = O.Rolling(bl[b], ps[b], init, child)
        e = O.Env(cs, N, "C+P+S-lb-soft", "diff")
        ok = True
        for t in range(N - child):
            rc, st, dy, nodes = ro.convert_to_input()
            ok &= rc == 0 and np.array_equal(st, seen[t][0][b]) and np.array_equal(dy, seen[t][1][b])
            ok &= np.array_equal(O.initial_mask(dy[None], child)[0], seen[t][2][b])
            p = int(tour[b, t])
            ok &= picked[b, t] == nodes[p % child]
            e.add_new_block(st[1:, p]); ro.remove(p % child)
        rc, st, dy, nodes = ro.convert_to_input()
        ok &= rc == 1 and np.array_equal(st, seen[N - child][0][b]) and np.array_equal(dy, seen[N - child][1][b])
        for t in range(N - child, N):
            e.add_new_block(st[1:, int(tour[b, t])])
        ok &= sorted(picked[b].tolist()) == list(range(N))
        ok &= bool(np.float32(e.calc_ratio()) == -out["reward"][b].item())
        bad += not ok
    return bad


t0 = time.time(); nbad = 0; windows = 0; cases = []
n_cfg = int(sys.argv[1])
for k in range(n_cfg):
    rs = np.random.RandomState(9000 + k)
    D = int(rs.choice([2, 3]))
    big = rs.rand() < 0.15
    child = int(rs.randint(2, 25)) if not big else int(rs.randint(2, 40))
    N = int(rs.randint(child + 1, 65)) if not big else int(rs.randint(max(65, child + 1), 330))    # round 5: up to 256 on one wavefront, thread-per-instance above
    W = int(rs.randint(4, 9))
    B = 48 if not big else 16
    fused = bool(rs.rand() < 0.6)
    b = one(D, N, child, W, B, 100 + k, fused)
    windows += (N - child + 1) * B
    if b:
        nbad += b; cases.append((D, N, child, W, fused, b)); print("CASE", D, N, child, W, fused, "bad instances", b, flush=True)
summary = dict(script="scripts/stress_rolling.py", configurations=n_cfg, windows=windows, mismatching_instances=nbad,
               seconds=round(time.time() - t0, 1), cases=cases,
               compared="static, dynamic, node list, initial mask of every window; final packing ratio; for a slice of each batch")
print(json.dumps(summary))
if len(sys.argv) > 2:
    json.dump(summary, open(sys.argv[2], "w"), indent=1)
