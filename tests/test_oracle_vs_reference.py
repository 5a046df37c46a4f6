"""Live differential test: CPU oracle vs the imported reference (build container only).

Skipped wherever /root/reference is absent (e.g. the GPU box).  Compares the *full* state after
every step, including the voxel grid.
"""
import numpy as np
import pytest

import oracle_lib as O
import ref_loader

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not ref_loader.available(), reason="reference checkout not present")]


@pytest.fixture(scope="module")
def ref():
    return ref_loader.load()


def _diff(tools, cs, n, reward, feat, strategy, episodes, seed, lo=1, hi=5, hz=None):
    rng = np.random.RandomState(seed)
    D = len(cs)
    for ep in range(episodes):
        blocks = rng.randint(lo, hi, size=(n, D))
        if hz is not None:                                   # wide blocks, heights below hz
            blocks[:, -1] = rng.randint(1, hz, size=n)
        r = tools.Container(list(cs), n, reward, feat, packing_strategy=strategy)
        o = O.Env(cs, n, reward, feat, strategy)
        for t in range(n):
            rf = r.add_new_block(blocks[t].astype(np.float32))
            rc, of = o.add_new_block(blocks[t])
            ctx = (cs, reward, strategy, ep, t, blocks[: t + 1].tolist())
            assert rc == 0, ctx
            assert np.array_equal(np.asarray(rf), of), ctx
            assert np.array_equal(r.heightmap, o.heightmap), ctx
            assert np.array_equal(r.positions, o.positions), ctx
            assert list(r.stable) == o.stable.tolist(), ctx
            assert int(r.valid_size) == o.valid_size and int(r.empty_size) == o.empty_size, ctx
            assert np.array_equal(r.container, o.container), ctx
        a, b = float(r.calc_ratio()), o.calc_ratio()
        assert a == b or (np.isnan(a) and np.isnan(b))


@pytest.mark.parametrize("reward", ["C+P+S-lb-soft", "C+P+S-lb-hard", "C+P-lb-soft", "C+P-lb-hard"])
def test_lbg2d(ref, reward):
    _diff(ref[0], [5, 50], 10, reward, "diff", "LB_GREEDY", 60, 11)
    _diff(ref[0], [7, 100], 20, reward, "zero", "LB_GREEDY", 20, 12)
    _diff(ref[0], [3, 60], 8, reward, "full", "LB_GREEDY", 20, 13, 1, 6)


@pytest.mark.parametrize("reward", ["C+P+S-lb-soft", "C+P+S-lb-hard"])
def test_lbg3d(ref, reward):
    _diff(ref[0], [5, 5, 50], 10, reward, "diff", "LB_GREEDY", 40, 21)
    _diff(ref[0], [5, 5, 250], 30, reward, "full", "LB_GREEDY", 6, 22)
    _diff(ref[0], [6, 6, 80], 12, reward, "diff", "LB_GREEDY", 12, 23, 1, 7)


@pytest.mark.parametrize("reward", ["C+P+S-lb-soft", "C+P+S-lb-hard", "C+P-lb-soft"])
def test_lb_legacy(ref, reward):
    """packing_strategy 'LB' (tools.py:3683-3686, :1602-1955), full state incl. the voxel grid."""
    _diff(ref[0], [5, 50], 10, reward, "diff", "LB", 40, 51)
    _diff(ref[0], [7, 100], 20, reward, "zero", "LB", 10, 52)
    _diff(ref[0], [3, 60], 8, reward, "full", "LB", 15, 53, 1, 6)
    _diff(ref[0], [5, 5, 50], 10, reward, "diff", "LB", 20, 54)
    _diff(ref[0], [4, 6, 80], 12, reward, "full", "LB", 6, 55, 1, 6)


@pytest.mark.parametrize("reward", ["C+P+S-mcs-soft", "C+P+S-mcs-hard", "C+P+S-mul-soft", "C+P-mcs-soft"])
def test_macs2d(ref, reward):
    _diff(ref[0], [7, 100], 20, reward, "diff", "MACS", 25, 31)
    _diff(ref[0], [5, 50], 10, reward, "diff", "MACS", 40, 32)
    # wide containers (the kernels' 32- and 64-lane form, tap_macs_wide.h), blocks up to 8 wide
    _diff(ref[0], [20, 60], 24, reward, "diff", "MACS", 4, 33, 1, 7)
    _diff(ref[0], [40, 40], 30, reward, "full", "MACS", 2, 34, 1, 9)
    _diff(ref[0], [64, 30], 24, reward, "zero", "MACS", 2, 35, 1, 9)
    # above 64 columns (the kernels' one-thread-per-container form, macs_big.hip), blocks up to 70 wide
    _diff(ref[0], [100, 40], 30, reward, "diff", "MACS", 2, 36, 1, 14)
    _diff(ref[0], [70, 60], 16, reward, "full", "MACS", 2, 37, 1, 30, hz=7)
    _diff(ref[0], [130, 50], 12, reward, "zero", "MACS", 1, 38, 2, 71, hz=6)


@pytest.mark.parametrize("reward", ["C+P+S-mcs-soft", "C+P+S-mcs-hard", "C+P+S-mul-soft", "mcs-soft"])
def test_macs3d(ref, reward):
    _diff(ref[0], [5, 5, 50], 10, reward, "diff", "MACS", 12, 41)
    _diff(ref[0], [6, 6, 60], 16, reward, "full", "MACS", 4, 42, 1, 6)
    _diff(ref[0], [4, 7, 40], 12, reward, "diff", "MACS", 4, 43)
    # above 64 cells / sides above 8 (the kernels' one-thread-per-container form, macs3_big.hip), footprints up to 8 x 8
    _diff(ref[0], [10, 10, 40], 24, reward, "diff", "MACS", 2, 44, 1, 6)
    _diff(ref[0], [12, 9, 30], 20, reward, "full", "MACS", 1, 45, 1, 9, hz=5)
    _diff(ref[0], [9, 16, 30], 20, reward, "zero", "MACS", 1, 46, 2, 8, hz=6)


@pytest.mark.parametrize("shape", [(32, 6, 2), (16, 30, 2), (8, 42, 6), (8, 22, 6)])   # 18 .. 126 rows
def test_masks(ref, shape):
    import torch
    pack = ref[1]
    rng = np.random.RandomState(5)
    B, n, R = shape
    dyn = (rng.rand(B, 3 * n, n * R) < 0.12).astype(np.float32)
    static = np.zeros((B, 3 if R == 2 else 4, n * R), np.float32)     # 1 + D rows: pack.py reads D off the shape
    static[:, 0, :] = np.tile(np.arange(n), R)
    mask = (rng.rand(B, n * R) < 0.8).astype(np.float32)
    ptr = rng.randint(0, n * R, size=B).astype(np.int64)
    rd = pack.update_dynamic(torch.from_numpy(dyn), torch.from_numpy(static), torch.from_numpy(ptr), "bot", True)
    rc, rm = pack.update_mask(torch.from_numpy(mask), rd, torch.from_numpy(static), torch.from_numpy(ptr), "bot", True)
    od = O.update_dynamic(dyn, static, ptr, n, 3)
    oc, om = O.update_mask(mask, od, ptr, n, R)
    assert np.array_equal(rd.numpy(), od) and np.array_equal(rc.numpy(), oc) and np.array_equal(rm.numpy(), om)


@pytest.mark.parametrize("D", [2, 3])
def test_reward_mul(ref, D):
    """pack.reward for the two-container input type: blocks split by target id, packed separately."""
    import torch
    pack = ref[1]
    rng = np.random.RandomState(17 + D)
    B, n = 24, 8
    R = 2 if D == 2 else 6
    blocks = rng.randint(1, 5, size=(B, D, n)).astype(np.float32)
    static = np.zeros((B, 1 + D + 1, n * R), np.float32)
    for r in range(R):
        static[:, 0, r * n:(r + 1) * n] = np.arange(n)
        static[:, 1:1 + D, r * n:(r + 1) * n] = blocks          # same sides in every rotation slot: enough here
        static[:, -1, r * n:(r + 1) * n] = rng.randint(0, 2, size=(B, n))
    static[0, -1, :] = 0                                         # one env with an empty second container
    tour = np.stack([rng.permutation(n * R)[:n] for _ in range(B)]).astype(np.int64)
    want = pack.reward(torch.from_numpy(static), torch.from_numpy(tour), "C+P+S-lb-soft", "mul", True, 5, 60).numpy()
    got = O.reward_mul(static, tour, "C+P+S-lb-soft", 5, 60, R)
    assert np.array_equal(got, want)
