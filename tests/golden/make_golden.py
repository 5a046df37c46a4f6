#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the UPSTREAM REFERENCE itself.

Runs only in the build container (needs /root/reference); the GPU box gets the committed
.npz files, never the reference.  Every fixture holds inputs and the reference's outputs --
no reference source text.  Usage:

    python tests/golden/make_golden.py [--only NAME ...]

Fixtures:
  lbg2d.npz, lbg3d.npz, macs2d.npz, macs3d.npz, lb_legacy.npz   tools.Container.add_new_block / calc_ratio step traces
  stable3d.npz                       tools.is_stable, exhaustive over footprints <= 4x4 (+5xk samples)
  stable3d_wide.npz                  tools.is_stable, sampled, footprints with a side of 9 .. 16
  dataset_2d.npz, dataset_3d.npz     pack.create_dataset -> text files -> pack.PACKDataset tensors
  masks_2d.npz, masks_3d.npz         pack.update_dynamic / pack.update_mask traces on random feasible tapes
  episode_2d.npz, episode_3d.npz     reference DRL.forward (pretrained actor, greedy) traces
  reward_tour.npz                    pack.reward on those tours
  kat.npz                            hand-checkable known answers
  rolling.npz                        generate.InitialContainer window traces (rolling.py's outer loop)
  rolling_big.npz                    the same for instances of 70 .. 300 blocks (node ids above 63, above 255)
  dataset_wide.npz                   pack.create_dataset with initial containers of 100 / 144 cells (3D) and 70 columns (2D)
  render.npz                         pack.render's eight metric files (LB_GREEDY / MACS / MUL, 2D / 3D, two-container types)
  ppsg3d.npz                         generate.BPP_Generator_3D / generate_blocks_with_GT (3D) under recorded seeds
  ppsg2d.npz                         generate.BPP_Generator_2D_easy / generate_blocks_with_GT (2D) under recorded seeds
  ppsg2d_arm.npz                     generate_blocks_with_GT / generate_blocks (2D) with arm_size 2 .. 4 under recorded seeds
  ppsg_2d.npz                        (--only ppsg) 64 instances of the reference's PPSG generator + MACS traces over them
"""
import argparse
import itertools
import math
import os
import sys
import tempfile
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ref_loader  # noqa: E402

warnings.filterwarnings("ignore")


def rand_blocks(seed, B, n, D, lo=1, hi=5, marginal=True):
    """RAND marginal: each side iid in {1..4}, p = [.15,.35,.35,.15] (generate.py:881-882,896)."""
    rng = np.random.RandomState(seed)
    if marginal and (lo, hi) == (1, 5):
        return rng.choice([1, 2, 3, 4], size=(B, n, D), p=[0.15, 0.35, 0.35, 0.15]).astype(np.int8)
    return rng.randint(lo, hi, size=(B, n, D)).astype(np.int8)


def trace_container(tools, cs, n, reward, feat, strategy, blocks):
    """Run B reference Containers step by step; return per-step traces."""
    B = blocks.shape[0]
    D = len(cs)
    cells = int(np.prod(cs[:-1]))
    feats, hms, poss, stabs, valids, emptys, ratios, cpss = [], [], [], [], [], [], [], []
    for b in range(B):
        c = tools.Container(list(cs), n, reward, feat, packing_strategy=strategy)
        f_b, h_b, v_b, e_b = [], [], [], []
        for t in range(n):
            f = c.add_new_block(blocks[b, t].astype(np.float32), False)
            f_b.append(np.asarray(f).reshape(-1).copy())
            h_b.append(np.asarray(c.heightmap).reshape(-1).copy())
            v_b.append(int(c.valid_size))
            e_b.append(int(c.empty_size))
        feats.append(f_b); hms.append(h_b); valids.append(v_b); emptys.append(e_b)
        poss.append(np.asarray(c.positions).copy())
        stabs.append(np.asarray(c.stable, dtype=bool).copy())
        ratios.append(float(c.calc_ratio()))
        cpss.append([float(x) for x in c.calc_CPS()])
    return dict(
        features=np.asarray(feats, dtype=np.int16).reshape(B, n, -1),
        heightmaps=np.asarray(hms, dtype=np.int16).reshape(B, n, cells),
        positions=np.asarray(poss, dtype=np.int16).reshape(B, n, D),
        stable=np.asarray(stabs, dtype=np.uint8),
        valid=np.asarray(valids, dtype=np.int32), empty=np.asarray(emptys, dtype=np.int32),
        ratio=np.asarray(ratios, dtype=np.float64), cps=np.asarray(cpss, dtype=np.float64))


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print("wrote %-18s %7.1f KB" % (name, os.path.getsize(path) / 1024.0))


def pack_cases(cases):
    """cases: list of (meta dict, trace dict, blocks) -> flat npz dict with a `cases` index."""
    out = {}
    metas = []
    for i, (meta, tr, blocks) in enumerate(cases):
        metas.append(repr(meta))
        out["c%d_blocks" % i] = blocks
        for k, v in tr.items():
            out["c%d_%s" % (i, k)] = v
    out["cases"] = np.asarray(metas)
    return out


# ----------------------------------------------------------------------------------------------

def make_lbg2d(tools):
    cases = []
    for reward in ("C+P+S-lb-soft", "C+P+S-lb-hard", "C+P-lb-soft", "C+P-lb-hard"):
        blocks = rand_blocks(101, 256, 10, 2)
        meta = dict(cs=[5, 50], n=10, reward=reward, feat="diff", strategy="LB_GREEDY")
        cases.append((meta, trace_container(tools, [5, 50], 10, reward, "diff", "LB_GREEDY", blocks), blocks))
    # widths 1..8, sides 1..5 (includes blocks wider than the container), all three feature types
    for W in range(1, 9):
        for reward, feat in (("C+P+S-lb-soft", "diff"), ("C+P+S-lb-hard", "zero"), ("C+P-lb-soft", "full")):
            blocks = rand_blocks(200 + W, 24, 8, 2, 1, 6, marginal=False)
            meta = dict(cs=[W, 60], n=8, reward=reward, feat=feat, strategy="LB_GREEDY")
            cases.append((meta, trace_container(tools, [W, 60], 8, reward, feat, "LB_GREEDY", blocks), blocks))
    # config-4 shape driven by LB_GREEDY, n=20 W=7
    blocks = rand_blocks(103, 64, 20, 2)
    meta = dict(cs=[7, 100], n=20, reward="C+P+S-lb-soft", feat="diff", strategy="LB_GREEDY")
    cases.append((meta, trace_container(tools, [7, 100], 20, "C+P+S-lb-soft", "diff", "LB_GREEDY", blocks), blocks))
    save("lbg2d.npz", **pack_cases(cases))


def make_lbg3d(tools):
    cases = []
    for reward in ("C+P+S-lb-soft", "C+P+S-lb-hard", "C+P-lb-soft"):
        blocks = rand_blocks(301, 192, 10, 3)
        meta = dict(cs=[5, 5, 50], n=10, reward=reward, feat="diff", strategy="LB_GREEDY")
        cases.append((meta, trace_container(tools, [5, 5, 50], 10, reward, "diff", "LB_GREEDY", blocks), blocks))
    # config-5 shape: 50 placements, H = 250
    blocks = rand_blocks(302, 24, 50, 3)
    meta = dict(cs=[5, 5, 250], n=50, reward="C+P+S-lb-soft", feat="diff", strategy="LB_GREEDY")
    cases.append((meta, trace_container(tools, [5, 5, 250], 50, "C+P+S-lb-soft", "diff", "LB_GREEDY", blocks), blocks))
    # other widths, bigger sides (footprints up to 6x6), other feature types
    for W, hi, reward, feat in ((3, 4, "C+P+S-lb-soft", "full"), (4, 5, "C+P+S-lb-hard", "zero"),
                                (6, 7, "C+P+S-lb-soft", "diff"), (7, 6, "C+P+S-lb-hard", "diff"),
                                (8, 5, "C+P+S-lb-soft", "diff")):
        blocks = rand_blocks(310 + W, 24, 12, 3, 1, hi, marginal=False)
        meta = dict(cs=[W, W, 90], n=12, reward=reward, feat=feat, strategy="LB_GREEDY")
        cases.append((meta, trace_container(tools, [W, W, 90], 12, reward, feat, "LB_GREEDY", blocks), blocks))
    save("lbg3d.npz", **pack_cases(cases))


def make_lb_legacy(tools):
    """packing_strategy 'LB' (tools.py:3683-3686 -> calc_one_position_greedy_2d / _3d, :1602-1955): step traces
    of the reference Container, 2D and 3D, soft and hard, several widths and feature types."""
    cases = []
    for reward in ("C+P+S-lb-soft", "C+P+S-lb-hard", "C+P-lb-soft", "C+P-lb-hard"):
        blocks = rand_blocks(501, 128, 10, 2)
        meta = dict(cs=[5, 50], n=10, reward=reward, feat="diff", strategy="LB")
        cases.append((meta, trace_container(tools, [5, 50], 10, reward, "diff", "LB", blocks), blocks))
    for W in range(1, 9):
        for reward, feat in (("C+P+S-lb-soft", "diff"), ("C+P+S-lb-hard", "zero"), ("C+P-lb-soft", "full")):
            blocks = rand_blocks(520 + W, 16, 8, 2, 1, 6, marginal=False)
            meta = dict(cs=[W, 60], n=8, reward=reward, feat=feat, strategy="LB")
            cases.append((meta, trace_container(tools, [W, 60], 8, reward, feat, "LB", blocks), blocks))
    blocks = rand_blocks(503, 32, 20, 2)
    meta = dict(cs=[7, 100], n=20, reward="C+P+S-lb-soft", feat="diff", strategy="LB")
    cases.append((meta, trace_container(tools, [7, 100], 20, "C+P+S-lb-soft", "diff", "LB", blocks), blocks))
    for reward in ("C+P+S-lb-soft", "C+P+S-lb-hard", "C+P-lb-soft"):
        blocks = rand_blocks(541, 96, 10, 3)
        meta = dict(cs=[5, 5, 50], n=10, reward=reward, feat="diff", strategy="LB")
        cases.append((meta, trace_container(tools, [5, 5, 50], 10, reward, "diff", "LB", blocks), blocks))
    for W, L, hi, reward, feat in ((3, 3, 4, "C+P+S-lb-soft", "full"), (4, 6, 5, "C+P+S-lb-hard", "zero"),
                                   (6, 4, 6, "C+P+S-lb-soft", "diff"), (7, 7, 5, "C+P+S-lb-hard", "diff")):
        blocks = rand_blocks(550 + W, 16, 12, 3, 1, hi, marginal=False)
        meta = dict(cs=[W, L, 90], n=12, reward=reward, feat=feat, strategy="LB")
        cases.append((meta, trace_container(tools, [W, L, 90], 12, reward, feat, "LB", blocks), blocks))
    blocks = rand_blocks(560, 12, 30, 3)
    meta = dict(cs=[5, 5, 150], n=30, reward="C+P+S-lb-soft", feat="diff", strategy="LB")
    cases.append((meta, trace_container(tools, [5, 5, 150], 30, "C+P+S-lb-soft", "diff", "LB", blocks), blocks))
    save("lb_legacy.npz", **pack_cases(cases))


def make_macs2d(tools):
    cases = []
    for reward in ("C+P+S-mcs-soft", "C+P+S-mcs-hard", "C+P+S-lb-soft", "C+P+S-mul-soft",
                   "C+P+S-mul-hard", "C+P-mcs-soft"):
        blocks = rand_blocks(401, 96, 20, 2)
        meta = dict(cs=[7, 100], n=20, reward=reward, feat="diff", strategy="MACS")
        cases.append((meta, trace_container(tools, [7, 100], 20, reward, "diff", "MACS", blocks), blocks))
    for W in (3, 4, 5, 6, 8):
        for reward in ("C+P+S-mcs-soft", "C+P+S-mcs-hard"):
            # sides <= W: the reference raises IndexError (tools.py:2550) once a block wider than
            # the container has failed and a later step inspects its "top"
            blocks = rand_blocks(410 + W, 32, 10, 2, 1, min(6, W + 1), marginal=False)
            meta = dict(cs=[W, 60], n=10, reward=reward, feat="diff", strategy="MACS")
            cases.append((meta, trace_container(tools, [W, 60], 10, reward, "diff", "MACS", blocks), blocks))
    save("macs2d.npz", **pack_cases(cases))


def make_macs3d(tools):
    """MACS in 3D (tools.calc_one_position_mcs_3d, tools.py:2751-3165)."""
    cases = []
    for reward in ("C+P+S-mcs-soft", "C+P+S-mcs-hard", "C+P+S-mul-soft", "C+P+S-mul-hard", "mcs-soft"):
        blocks = rand_blocks(501, 24, 10, 3)
        meta = dict(cs=[5, 5, 50], n=10, reward=reward, feat="diff", strategy="MACS")
        cases.append((meta, trace_container(tools, [5, 5, 50], 10, reward, "diff", "MACS", blocks), blocks))
    for cs, n, hi, seed in (([6, 6, 60], 16, 6, 511), ([4, 7, 40], 12, 5, 512), ([7, 4, 40], 12, 5, 513),
                            ([8, 8, 64], 24, 5, 514)):
        for reward in ("C+P+S-mcs-soft", "C+P+S-mcs-hard"):
            blocks = rand_blocks(seed, 8, n, 3, 1, hi, marginal=False)
            meta = dict(cs=cs, n=n, reward=reward, feat="full", strategy="MACS")
            cases.append((meta, trace_container(tools, cs, n, reward, "full", "MACS", blocks), blocks))
    save("macs3d.npz", **pack_cases(cases))


def make_stable3d(tools):
    """tools.is_stable over every support mask of every footprint <= 4x4, plus samples for 5xk / kx5 / 6x6."""
    shapes, offs, bits = [], [], []
    rng = np.random.RandomState(7)
    total = 0
    sample_masks = {}
    for bx, by in itertools.product(range(1, 7), range(1, 7)):
        cells = bx * by
        if bx <= 4 and by <= 4:
            masks = np.arange(1 << cells, dtype=np.int64)
        elif cells <= 36 and (bx <= 6 and by <= 6):
            masks = rng.randint(0, 1 << min(cells, 62), size=1500, dtype=np.int64)
            sample_masks["m_%d_%d" % (bx, by)] = masks
        res = np.zeros(len(masks), np.uint8)
        cont = np.zeros((bx, by, 3), dtype=int)
        for mi, m in enumerate(masks):
            # bit (i*by + j) of m = voxel under footprint cell (i, j) is occupied
            lay = np.array([(int(m) >> k) & 1 for k in range(cells)], dtype=int).reshape(bx, by)
            cont[:, :, 0] = lay
            res[mi] = bool(tools.is_stable(np.array([bx, by, 1]), np.array([0, 0, 1]), cont))
        shapes.append((bx, by)); offs.append(total); total += len(masks)
        bits.append(res)
    allbits = np.concatenate(bits)
    save("stable3d.npz", shapes=np.asarray(shapes, np.int8), offsets=np.asarray(offs, np.int64),
         bits=np.packbits(allbits), count=np.int64(total), **sample_masks)


WIDE_FOOTPRINTS = ((9, 9), (10, 10), (12, 7), (16, 16), (3, 16), (16, 2), (10, 4), (9, 1), (13, 13), (2, 11))


def make_stable3d_wide(tools):
    """tools.is_stable for footprints with a side above 8 (up to 16 x 16: the device's tap_stable_wide.h), sampled:
    400 support patterns per footprint with 2 .. cells/2 supported cells (more is the trivial majority case), a third
    of them confined to one or two rows / columns so that the collinear fall-back (qhull raising) is well covered.
    Stored as one byte per footprint cell, row-major (i outer, j inner)."""
    rng = np.random.RandomState(17)
    out, shapes = {}, []
    for si, (bx, by) in enumerate(WIDE_FOOTPRINTS):
        cells = bx * by
        masks, res = np.zeros((400, cells), np.uint8), np.zeros(400, np.uint8)
        cont = np.zeros((bx, by, 3), dtype=int)
        for mi in range(400):
            k = int(rng.randint(2, max(3, cells // 2 + 2)))
            lay = np.zeros((bx, by), int)
            kind = mi % 6
            if kind == 0 and by > 1:                       # one column of constant j (collinear)
                lay[rng.choice(bx, size=min(k, bx), replace=False), rng.randint(by)] = 1
            elif kind == 1 and bx > 1:                     # one row of constant i (collinear, p0 == p1 quirk)
                lay[rng.randint(bx), rng.choice(by, size=min(k, by), replace=False)] = 1
            elif kind == 2 and min(bx, by) > 2:            # a diagonal
                m = min(bx, by)
                idx = rng.choice(m, size=min(k, m), replace=False)
                lay[idx, idx] = 1
            else:
                lay.reshape(-1)[rng.choice(cells, size=min(k, cells), replace=False)] = 1
            cont[:, :, 0] = lay
            masks[mi] = lay.reshape(-1)
            res[mi] = bool(tools.is_stable(np.array([bx, by, 1]), np.array([0, 0, 1]), cont))
        shapes.append((bx, by))
        out["m%d" % si] = np.packbits(masks, axis=1)
        out["r%d" % si] = res
    save("stable3d_wide.npz", shapes=np.asarray(shapes, np.int8), **out)


def _ref_dataset(pack, D, n, count, seed, tmp):
    """pack.create_dataset -> (dir, raw text lines) for `count` validation samples."""
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        _, valid_dir = pack.create_dataset(n, 2, count, D, 7, 50, 1, [1, 5], seed=seed)
        valid_dir = os.path.abspath(valid_dir) + "/"
    finally:
        os.chdir(cwd)
    return valid_dir


def make_dataset(pack, D, tmp, count):
    import torch
    n = 10
    vdir = _ref_dataset(pack, D, n, count, 12345, tmp)
    ds = pack.PACKDataset(vdir, n, count, 12345, "bot", "diff", True, 5, unit=1)
    files = {}
    for f in ("blocks", "pos", "container", "dep_move", "dep_small", "dep_large"):
        files["txt_" + f] = np.loadtxt(vdir + f + ".txt").astype(np.int8)
    static = ds.static.detach().numpy().astype(np.int8)        # small ints stored as floats
    dynamic = ds.dynamic.detach().numpy().astype(np.int8)
    assert np.array_equal(static.astype(np.float32), ds.static.detach().numpy())
    assert np.array_equal(dynamic.astype(np.float32), ds.dynamic.detach().numpy())
    save("dataset_%dd.npz" % D, static=static, dynamic=dynamic,
         decoder_static_shape=np.asarray(ds.decoder_static.shape),
         decoder_dynamic_shape=np.asarray(ds.decoder_dynamic.shape), **files)
    return torch.from_numpy(static.astype(np.float32)), torch.from_numpy(dynamic.astype(np.float32))


WIDE_DATASETS = ((3, 10, 10, 40, 24), (2, 12, 70, 30, 24), (3, 12, 12, 40, 8))   # D, n, initial width, height, samples


def make_dataset_wide(pack):
    """pack.create_dataset with --initial_container_width beyond the lane-per-cell kernels (3D 10 x 10 and 12 x 12: 100
    and 144 cells; 2D 70 columns): generate.generate_blocks packs into the initial container with
    tools.calc_positions_lb_greedy in hard mode (generate.py:908) for any width.  Stored: the text files' blocks and
    positions and the PACKDataset tensors (dataset_wide.npz, round 5)."""
    out, cases = {}, []
    for D, n, W, H, count in WIDE_DATASETS:
        with tempfile.TemporaryDirectory() as tmp:
            cwd = os.getcwd()
            os.chdir(tmp)
            try:
                _, vdir = pack.create_dataset(n, 2, count, D, W, H, 1, [1, 5], seed=4321 + W)
                vdir = os.path.abspath(vdir) + "/"
            finally:
                os.chdir(cwd)
            ds = pack.PACKDataset(vdir, n, count, 4321 + W, "bot", "diff", True, 5, unit=1)
            tag = "w%d_" % len(cases)
            cases.append(repr(dict(D=D, n=n, W=W, H=H, count=count)))
            out[tag + "txt_blocks"] = np.loadtxt(vdir + "blocks.txt").astype(np.int8)
            out[tag + "txt_pos"] = np.loadtxt(vdir + "pos.txt").astype(np.int16)
            out[tag + "static"] = ds.static.detach().numpy().astype(np.int8)
            out[tag + "dynamic"] = ds.dynamic.detach().numpy().astype(np.int8)
            assert np.array_equal(out[tag + "static"].astype(np.float32), ds.static.detach().numpy())
    out["cases"] = np.asarray(cases)
    save("dataset_wide.npz", **out)


def make_ppsg(tools, pack, src=None):
    """64 PPSG instances (2D, 20 blocks, initial container 7 wide) written by the reference's own
    generator -- SURVEY 8(c)/(d): pack.create_dataset(10, 8, 10000, 2, 7, 50, 1, [1, 5], seed=12345)
    (generate_height_prob reads its 10 000 validation samples) followed by
    pack.create_dataset_gt(20, 8, 64, 2, 7, 50, 7, 50, 'bot', 1, [1, 5], seed=12345).  The
    perfect-packing generator is a rejection sampler and takes the better part of an hour, so an
    already generated directory can be passed with --ppsg-dir.  Stored: the PACKDataset tensors and the
    MACS trace of the reference container over the blocks in file order (rotation 0)."""
    n = 20
    if src is None:
        tmp = tempfile.mkdtemp()
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            pack.create_dataset(10, 8, 10000, 2, 7, 50, 1, [1, 5], seed=12345)
            _, src = pack.create_dataset_gt(n, 8, 64, 2, 7, 50, 7, 50, "bot", 1, [1, 5], seed=12345)
            src = os.path.abspath(src)
        finally:
            os.chdir(cwd)
    src = src.rstrip("/") + "/"
    ds = pack.PACKDataset(src, n, 64, 12345, "bot", "diff", True, 7, unit=1)
    static = ds.static.detach().numpy().astype(np.int8)
    dynamic = ds.dynamic.detach().numpy().astype(np.int8)
    assert np.array_equal(static.astype(np.float32), ds.static.detach().numpy())
    assert np.array_equal(dynamic.astype(np.float32), ds.dynamic.detach().numpy())
    blocks = np.ascontiguousarray(static[:, 1:, :n].transpose(0, 2, 1)).astype(np.int8)   # rotation 0, file order
    cases = []
    for reward in ("C+P+S-mcs-soft", "C+P+S-mcs-hard"):
        meta = dict(cs=[7, 100], n=n, reward=reward, feat="diff", strategy="MACS")
        cases.append((meta, trace_container(tools, [7, 100], n, reward, "diff", "MACS", blocks), blocks))
    files = {}
    for f in ("blocks", "pos", "container", "dep_move", "dep_small", "dep_large"):
        files["txt_" + f] = np.loadtxt(src + f + ".txt").astype(np.int8)
    save("ppsg_2d.npz", static=static, dynamic=dynamic, **files, **pack_cases(cases))


def make_ppsg3d(pack, generate):
    """The reference's 3D perfect-packing generator under recorded seeds:

    * `bpp_*`: single calls of generate.BPP_Generator_3D(n, gt_size, [1, 5]) after np.random.seed(seed), for a
      spread of (n, gt_size), accepted or not -> blocks, positions;
    * `gt_*`: whole generate.generate_blocks_with_GT(n, gt_size, [7, 7, 50], 1, [1, 5], 'bot', 0) runs after
      np.random.seed(seed) -> its five return values, laid out as pack.create_dataset_gt writes them and read
      back through the reference's PACKDataset (static, dynamic), plus blocks / positions in layout order.

    The functions draw from numpy's global RandomState; a test re-creates the same MT19937 word stream with
    numpy (RandomState(seed)._bit_generator.random_raw) and feeds it to the oracle's restatement, which must
    then make the same decisions draw for draw.  Only seeds, parameters and outputs are stored."""
    out = {}
    bpp = []
    k = 0
    for n, gt in ((2, [5, 5, 1]), (3, [5, 5, 2]), (6, [5, 5, 4]), (10, [5, 5, 6]), (10, [5, 5, 7]), (12, [6, 4, 8]),
                  (20, [5, 5, 12]), (50, [5, 5, 31]), (64, [7, 7, 20])):
        for seed in range(40 if n <= 12 else 12):
            np.random.seed(1000 * n + seed)
            blocks, positions, _ = generate.BPP_Generator_3D(n, list(gt), [1, 5])
            out["bpp%d_blocks" % k] = blocks.astype(np.int16)
            out["bpp%d_positions" % k] = positions.astype(np.int16)
            bpp.append((n, gt[0], gt[1], gt[2], 1000 * n + seed))
            k += 1
    out["bpp_cases"] = np.asarray(bpp, dtype=np.int64)
    # accepted perfect packings by rejection, as generate_blocks_with_GT's first loop finds them
    full = []
    k = 0
    tmp = tempfile.mkdtemp()
    for n, gt in ((6, [5, 5, 4]), (8, [5, 5, 5]), (10, [5, 5, 6]), (10, [5, 5, 7])):
        for seed in range(6):
            sd = 77000 + 100 * n + 10 * gt[2] + seed
            np.random.seed(sd)
            rb, pos, dm, small, large = generate.generate_blocks_with_GT(n, list(gt), [7, 7, 50], 1, [1, 5], "bot", 0)
            d = os.path.join(tmp, "gt%d" % k) + "/"
            os.makedirs(d)
            with open(d + "blocks.txt", "w") as fb, open(d + "dep_small.txt", "w") as fs, open(d + "dep_large.txt", "w") as fl:
                for r in range(len(rb)):                       # pack.py:536-539
                    fb.write(" ".join(str(int(v)) for v in rb[r]) + "\n")
                    fs.write(" ".join(str(int(v)) for v in small[r]) + "\n")
                    fl.write(" ".join(str(int(v)) for v in large[r]) + "\n")
            open(d + "pos.txt", "w").write(" ".join(str(int(v)) for v in pos) + "\n")
            open(d + "dep_move.txt", "w").write(" ".join(str(int(v)) for v in dm) + "\n")
            open(d + "container.txt", "w").write(" ".join("0" for _ in range(n)) + "\n")
            ds = pack.PACKDataset(d, n, 1, 12345, "bot", "diff", True, 5, unit=1)
            out["gt%d_static" % k] = ds.static.detach().numpy()[0].astype(np.int8)
            out["gt%d_dynamic" % k] = ds.dynamic.detach().numpy()[0].astype(np.int8)
            out["gt%d_blocks" % k] = np.asarray(rb[0]).reshape(3, n).T.astype(np.int16)   # rotation 0, layout order
            out["gt%d_positions" % k] = np.asarray(pos).reshape(3, n).T.astype(np.int16)
            full.append((n, gt[0], gt[1], gt[2], sd))
            k += 1
            print("ppsg3d gt case", k, n, gt, flush=True)
    out["gt_cases"] = np.asarray(full, dtype=np.int64)
    save("ppsg3d.npz", **out)


def make_ppsg2d(generate):
    """The reference's 2D perfect-packing generator under recorded seeds (what generate_blocks_with_GT calls for
    block_dim 2, generate.py:72-73):

    * `bpp_*`: single calls of generate.BPP_Generator_2D_easy(n, [W, H], [1, 5]) after np.random.seed(seed) --
      volume-weighted block choice among the too-large (else the splittable) blocks, the axis rule with its
      position-for-size slip (:447), uniform and Gaussian split positions -- accepted or not;
    * `gt_*`: whole generate.generate_blocks_with_GT(n, [7, H], [7, 50], 1, [1, 5], 'bot', 0) runs -> blocks /
      positions in layout order (rotation 0).
    Seeds, parameters and outputs only; the oracle is driven by the same MT19937 word stream (ppsg3d above)."""
    out = {}
    bpp = []
    k = 0
    for n, gt in ((2, [5, 3]), (5, [7, 6]), (10, [7, 9]), (10, [7, 12]), (20, [7, 16]), (20, [7, 20]), (20, [7, 30]),
                  (30, [9, 40]), (50, [7, 60])):
        for seed in range(30):
            np.random.seed(5000 * n + seed)
            blocks, positions, _ = generate.BPP_Generator_2D_easy(n, list(gt), [1, 5])
            out["bpp%d_blocks" % k] = blocks.astype(np.int16)
            out["bpp%d_positions" % k] = positions.astype(np.int16)
            bpp.append((n, gt[0], gt[1], 5000 * n + seed))
            k += 1
    out["bpp_cases"] = np.asarray(bpp, dtype=np.int64)
    full = []
    k = 0
    for n, gt in ((6, [7, 5]), (8, [7, 7]), (10, [7, 9]), (12, [7, 10])):
        for seed in range(5):
            sd = 88000 + 100 * n + seed
            np.random.seed(sd)
            rb, pos, dm, small, large = generate.generate_blocks_with_GT(n, list(gt), [7, 50], 1, [1, 5], "bot", 0)
            out["gt%d_blocks" % k] = np.asarray(rb[0]).reshape(2, n).T.astype(np.int16)     # rotation 0, layout order
            out["gt%d_positions" % k] = np.asarray(pos).reshape(2, n).T.astype(np.int16)
            out["gt%d_dep_move" % k] = np.asarray(dm).astype(np.int8)
            full.append((n, gt[0], gt[1], sd))
            k += 1
            print("ppsg2d gt case", k, n, gt, flush=True)
    out["gt_cases"] = np.asarray(full, dtype=np.int64)
    save("ppsg2d.npz", **out)


def make_ppsg2d_arm(pack, generate):
    """generate.generate_blocks_with_GT (2D) and generate.generate_blocks with an arm wider than one column
    (`--arm_size`, generate.py:623-641): `arm_*` = whole generate_blocks_with_GT(n, [7, H], [W0, 50], arm, [1, 5], 'bot', 0)
    runs and `rnd_*` = generate_blocks(n, [W0, 50], arm, [1, 5]) runs after np.random.seed(seed), for arm_size 2 and 3 --
    the return values written as pack.create_dataset(_gt) writes them and read back through the reference's
    PACKDataset (static, dynamic), plus blocks / positions in layout order (rotation 0)."""
    out = {}
    tmp = tempfile.mkdtemp()

    def through_dataset(tag, n, rb, pos, dm, small, large):
        d = os.path.join(tmp, tag) + "/"
        os.makedirs(d)
        with open(d + "blocks.txt", "w") as fb, open(d + "dep_small.txt", "w") as fs, open(d + "dep_large.txt", "w") as fl:
            for r in range(len(rb)):                           # pack.py:536-539
                fb.write(" ".join(str(int(v)) for v in rb[r]) + "\n")
                fs.write(" ".join(str(int(v)) for v in small[r]) + "\n")
                fl.write(" ".join(str(int(v)) for v in large[r]) + "\n")
        open(d + "pos.txt", "w").write(" ".join(str(int(v)) for v in pos) + "\n")
        open(d + "dep_move.txt", "w").write(" ".join(str(int(v)) for v in dm) + "\n")
        open(d + "container.txt", "w").write(" ".join("0" for _ in range(n)) + "\n")
        ds = pack.PACKDataset(d, n, 1, 12345, "bot", "diff", True, 5, unit=1)
        out[tag + "_static"] = ds.static.detach().numpy()[0].astype(np.int8)
        out[tag + "_dynamic"] = ds.dynamic.detach().numpy()[0].astype(np.int8)
        out[tag + "_blocks"] = np.asarray(rb[0]).reshape(2, n).T.astype(np.int16)       # rotation 0, layout order
        out[tag + "_positions"] = np.asarray(pos).reshape(2, n).T.astype(np.int16)

    arm_cases, k = [], 0
    for n, gt, w0, arm in ((6, [7, 5], 7, 2), (8, [7, 7], 7, 2), (10, [7, 9], 9, 2), (10, [7, 9], 9, 3), (12, [7, 10], 10, 3)):
        for seed in range(4):
            sd = 99000 + 100 * n + 10 * arm + seed
            np.random.seed(sd)
            rb, pos, dm, small, large = generate.generate_blocks_with_GT(n, list(gt), [w0, 50], arm, [1, 5], "bot", 0)
            through_dataset("arm%d" % k, n, rb, pos, dm, small, large)
            arm_cases.append((n, gt[0], gt[1], w0, arm, sd))
            k += 1
            print("ppsg2d_arm gt case", k, n, gt, arm, flush=True)
    out["arm_cases"] = np.asarray(arm_cases, dtype=np.int64)
    rnd_cases, k = [], 0
    for n, w0, arm in ((10, 7, 2), (10, 9, 3), (20, 9, 2), (14, 12, 4)):
        for seed in range(6):
            sd = 66000 + 100 * n + 10 * arm + seed
            np.random.seed(sd)
            rb, pos, dm, small, large = generate.generate_blocks(n, [w0, 50 if n <= 14 else 90], arm, [1, 5])
            through_dataset("rnd%d" % k, n, rb, pos, dm, small, large)
            rnd_cases.append((n, w0, 50 if n <= 14 else 90, arm, sd))
            k += 1
    out["rnd_cases"] = np.asarray(rnd_cases, dtype=np.int64)
    save("ppsg2d_arm.npz", **out)


def make_masks(pack, D, static, dynamic):
    """Random feasible action tapes through the reference's update_dynamic / update_mask."""
    import torch
    B, rows, nR = dynamic.shape
    n = rows // 3
    R = nR // n
    rng = np.random.RandomState(55 + D)
    mask = torch.ones(B, nR)
    move = dynamic[:, :n].sum(1); small = dynamic[:, n:2 * n].sum(1); large = dynamic[:, 2 * n:].sum(1)
    cur = mask.clone()
    cur[(small * large + move).ne(0)] = 0.                      # model.py:297-307
    init_mask = cur.numpy().astype(np.uint8)
    dyn = dynamic
    ptrs, curs, masks, dyns = [], [], [], []
    for _ in range(n):
        c = cur.numpy()
        assert (c.sum(1) > 0).all(), "dead end in reference data"
        ptr = np.array([rng.choice(np.flatnonzero(c[b])) for b in range(B)], dtype=np.int64)
        ptr_t = torch.from_numpy(ptr)
        dyn = pack.update_dynamic(dyn, static, ptr_t, "bot", True)
        cur, mask = pack.update_mask(mask, dyn, static, ptr_t, "bot", True)
        ptrs.append(ptr); curs.append(cur.numpy().astype(np.uint8)); masks.append(mask.numpy().astype(np.uint8))
        dyns.append(np.packbits(dyn.numpy().astype(np.uint8), axis=None))
    assert not mask.byte().any()
    save("masks_%dd.npz" % D, ptr=np.asarray(ptrs, np.int16), initial_mask=init_mask,
         current_mask=np.asarray(curs), mask=np.asarray(masks), dynamic_bits=np.asarray(dyns),
         dyn_shape=np.asarray(dynamic.shape))


def make_episode(tools, pack, D, static, dynamic):
    """Reference DRL.forward, pretrained actor, greedy -- trace everything that crosses the seams."""
    import torch
    sys.path.insert(0, ref_loader.REFERENCE_DIR)
    import model as ref_model
    sys.path.remove(ref_loader.REFERENCE_DIR)
    n = 10
    rec = dict(feat=[], cur=[], mask=[], dyn=[])

    def upd(dynamic_, static_, ptr, input_type, allow_rot):
        out = pack.update_dynamic(dynamic_, static_, ptr, input_type, allow_rot)
        rec["dyn"].append(np.packbits(out.numpy().astype(np.uint8), axis=None))
        return out

    def msk(mask_, dynamic_, static_, ptr, input_type, allow_rot):
        c, m = pack.update_mask(mask_, dynamic_, static_, ptr, input_type, allow_rot)
        rec["cur"].append(c.numpy().astype(np.uint8)); rec["mask"].append(m.numpy().astype(np.uint8))
        return c, m

    orig_add = tools.Container.add_new_block

    def add(self, block, is_rotate=False):
        hm = orig_add(self, block, is_rotate)
        rec["feat"].append(np.asarray(hm).reshape(-1).copy())
        return hm

    tools.Container.add_new_block = add
    try:
        actor = ref_model.DRL(D, 3 * n, 128, 256, False, "bot", True, 5, 50, D, "C+P+S-lb-soft",
                              "shape_heightmap", "diff", "LB_GREEDY", upd, msk, 1, 0.1, 1.0)
        ck = os.path.join(ref_loader.REFERENCE_DIR, "pretrain_model",
                          "%dd-bot-C+P+S-lb-soft-width-5-note-sh-R-diff" % D, "actor.pt")
        actor.load_state_dict(torch.load(ck, map_location="cpu"))
        actor.eval()
        B = static.shape[0]
        dec_static = torch.zeros(B, D, 1)
        dec_dynamic = torch.zeros(B, 4, 1) if D == 2 else torch.zeros(B, 2, 5, 5)
        with torch.no_grad():
            tour_idx, tour_logp, _, neg_scores = actor(static, dynamic, [dec_static, dec_dynamic])
    finally:
        tools.Container.add_new_block = orig_add
    steps = tour_idx.shape[1]
    feats = np.asarray(rec["feat"], np.int16).reshape(steps, B, -1)
    save("episode_%dd.npz" % D, tour_idx=tour_idx.numpy().astype(np.int16), features=feats,
         current_mask=np.asarray(rec["cur"]), mask=np.asarray(rec["mask"]),
         dynamic_bits=np.asarray(rec["dyn"]), neg_scores=neg_scores.numpy().astype(np.float32),
         mean_reward=np.float64(neg_scores.mean().item()))
    return tour_idx


def make_reward_tour(pack, tours, statics):
    out = {}
    for D in (2, 3):
        r = pack.reward(statics[D], tours[D], "C+P+S-lb-soft", "bot", True, 5, 50, "LB_GREEDY")
        out["reward_%dd" % D] = r.numpy().astype(np.float32)
        # a second, non-trivial tape: reversed tour with swapped rotation
        out["tour_%dd" % D] = tours[D].numpy().astype(np.int16)
    save("reward_tour.npz", **out)


RENDER_SUFFIXES = ("ratio", "valid_size", "box_size", "empty_size", "stable_num", "packing_height", "time", "ids")


def make_render(pack):
    """pack.render (pack.py:670-807) -- what every `--task=test` run calls through validate (trainer.py:132, 493):
    the eight metric files it writes, for the committed data sets and the pretrained actors' tours, over
    LB_GREEDY / MACS / MUL, 2D and 3D, and the two-container input types.  Stored: the files' text."""
    import torch
    out, meta = {}, []
    rs = np.random.RandomState(77)
    for D in (2, 3):
        static = torch.from_numpy(np.load(os.path.join(HERE, "dataset_%dd.npz" % D))["static"].astype(np.float32))
        tour = torch.from_numpy(np.load(os.path.join(HERE, "episode_%dd.npz" % D))["tour_idx"].astype(np.int64))
        B, n = tour.shape
        ids = torch.from_numpy(rs.randint(0, 2, size=(B, 1, n)).astype(np.float32)).repeat(1, 1, static.shape[2] // n)
        ids[0] = 0.0                                   # one sample with an empty second container (pack.py:765)
        static_mul = torch.cat((static, ids), 1)
        out["target_ids_%dd" % D] = ids[:, 0, :n].numpy().astype(np.uint8)
        cases = [("bot", "LB_GREEDY", "C+P+S-lb-soft", 5, 50, 50), ("bot", "LB_GREEDY", "C+P-lb-hard", 5, 50, 50),
                 ("bot", "MACS", "C+P+S-mcs-soft", 5, 50, 50), ("bot", "MUL", "C+P+S-mul-hard", 5, 50, 50),
                 ("bot", "MACS", "mcs-soft", 5, 50, 50), ("bot", "MACS", "C+P-mcs-hard", 6, 60, 60),
                 ("mul", "LB_GREEDY", "C+P+S-lb-soft", 5, 50, 44), ("mul-with", "MACS", "C+P+S-mcs-soft", 5, 50, 44)]
        for k, (input_type, strategy, reward, W, H, H0) in enumerate(cases):
            cnt = B if D == 2 or strategy == "LB_GREEDY" else 24     # the reference's MACS 3D takes seconds per sample
            st = (static_mul if input_type.startswith("mul") else static)[:cnt]
            with tempfile.TemporaryDirectory() as tmp:
                path = os.path.join(tmp, "batch0_-1.2345.png")      # trainer.py:129-130
                pack.render(st, tour[:cnt], path, None, 0.125 + k, input_type=input_type, allow_rot=True,
                            container_width=W, container_height=H, initial_container_width=7,
                            initial_container_height=H0, unit=1.0, packing_strategy=strategy, reward_type=reward)
                names = sorted(os.listdir(tmp))
                assert names == sorted("batch-%s.txt" % s for s in RENDER_SUFFIXES), names
                for suf in RENDER_SUFFIXES:
                    with open(os.path.join(tmp, "batch-%s.txt" % suf), "rb") as f:
                        out["c%d_%dd_%s" % (k, D, suf)] = np.frombuffer(f.read(), dtype=np.uint8)
            meta.append("%d|%d|%s|%s|%s|%d|%d|%d|%d|%g" % (k, D, input_type, strategy, reward, W, H, H0, cnt, 0.125 + k))
    save("render.npz", meta=np.array(meta), **out)


def make_kat(tools):
    """Appendix-G style known answers, incl. calc_positions_lb_greedy's un-normalised ratio."""
    out = {}
    k2 = np.array([[3, 2], [1, 1], [1, 2], [2, 4], [5, 1]])
    pos, _, st, ratio, scores = tools.calc_positions_lb_greedy(k2, [5, 50], "C+P+S-lb-soft")
    out.update(k2_blocks=k2, k2_pos=pos, k2_stable=np.asarray(st), k2_ratio=np.float64(ratio),
               k2_scores=np.asarray(scores, np.int64))
    k3 = np.array([[3, 2, 2], [1, 1, 3], [1, 2, 1], [2, 4, 2], [5, 1, 1], [2, 2, 2]])
    pos, _, st, ratio, scores = tools.calc_positions_lb_greedy(k3, [5, 5, 50], "C+P+S-lb-soft")
    out.update(k3_blocks=k3, k3_pos=pos, k3_stable=np.asarray(st), k3_ratio=np.float64(ratio),
               k3_scores=np.asarray(scores, np.int64))
    k7 = np.array([[3, 2], [2, 1], [4, 1], [1, 3], [2, 2], [3, 1], [1, 1], [2, 3]])
    for tag, reward in (("soft", "C+P+S-lb-soft"), ("hard", "C+P+S-lb-hard"), ("cp", "C+P-lb-hard")):
        pos, _, st, ratio, scores = tools.calc_positions_lb_greedy(k7, [7, 50], reward)
        out.update({"k7_%s_pos" % tag: pos, "k7_%s_stable" % tag: np.asarray(st),
                    "k7_%s_ratio" % tag: np.float64(ratio), "k7_%s_scores" % tag: np.asarray(scores, np.int64)})
    out["k7_blocks"] = k7
    # a block that never fits: all-failed container -> nan ratio (0/0), counters untouched
    c = tools.Container([3, 20], 2, "C+P+S-lb-soft", "diff")
    f = c.add_new_block(np.array([4, 1], np.float32))
    out.update(fail_feature=np.asarray(f), fail_ratio=np.float64(c.calc_ratio()),
               fail_valid=np.int64(c.valid_size), fail_count=np.int64(c.current_blocks_num))
    save("kat.npz", **out)


ROLLING_SHAPES = ((2, 50, 10, [7, 250], 6), (3, 50, 10, [7, 7, 250], 4), (2, 24, 6, [6, 120], 6), (3, 20, 5, [5, 5, 100], 4))
# instances above 64 blocks (rolling.py:702 --total_blocks_num is free): node ids above 63 go through the CPython-set
# iteration order with a non-zero perturb, two-word graphs on the device (rolling_big.npz, round 3)
ROLLING_BIG_SHAPES = ((2, 100, 10, [7, 500], 2), (3, 100, 10, [7, 7, 500], 2), (2, 128, 16, [9, 600], 1), (3, 130, 12, [7, 7, 600], 1),
                      (2, 70, 20, [7, 400], 1),
                      # round 5: three- and four-word graphs (129 .. 256 blocks, one wavefront per instance) and the
                      # thread-per-instance form above 256 blocks
                      (3, 200, 10, [7, 7, 900], 1), (2, 300, 10, [9, 1400], 1))


def make_rolling(tools, generate, shapes=ROLLING_SHAPES, name="rolling.npz", seed=77):
    """generate.InitialContainer (generate.py:1589-1839) driven like rolling.validate
    (rolling.py:589-637): windows of 10 nodes over 50-/24-block instances, random feasible picks."""
    out = {}
    cases = []
    rng = np.random.RandomState(seed)
    for D, N, child, init, count in shapes:
        R = math.factorial(D)
        for c in range(count):
            np.random.seed(1000 * D + 10 * N + c)
            rot_blocks, positions, _, _, _ = generate.generate_blocks(N, list(init), 1, [1, 5])
            blocks_all = np.asarray(rot_blocks).reshape(R, D, N).transpose(0, 2, 1).reshape(R * N, D)  # rolling.py:484-486
            pos = np.asarray(positions).reshape(D, N).T                                               # rolling.py:491-492
            ic = generate.InitialContainer(blocks_all, pos, N, list(init), True, child, 'bot')
            statics, dynamics, nodes, ptrs = [], [], [], []
            while True:
                static, dynamic = ic.convert_to_input()
                statics.append(static.astype(np.int16)); dynamics.append(dynamic.astype(np.int8))
                nodes.append(np.asarray(ic.sub_graph_nodes, dtype=np.int16))
                if ic.is_last_graph():
                    break
                move = dynamic[:child].sum(0); small = dynamic[child:2 * child].sum(0); large = dynamic[2 * child:].sum(0)
                ok = np.flatnonzero((small * large + move) == 0)
                ptr = int(rng.choice(ok))
                ptrs.append(ptr)
                ic.remove_block(ic.sub_graph_nodes[ptr % child])
            tag = "r%d" % len(cases)
            cases.append(repr(dict(D=D, N=N, child=child, init=list(init))))
            out[tag + "_blocks"] = blocks_all[:N].astype(np.int8)
            out[tag + "_pos"] = pos.astype(np.int16)
            out[tag + "_static"] = np.asarray(statics)
            out[tag + "_dynamic"] = np.asarray(dynamics)
            out[tag + "_nodes"] = np.asarray(nodes)
            out[tag + "_ptr"] = np.asarray(ptrs, dtype=np.int16)
    out["cases"] = np.asarray(cases)
    save(name, **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--ppsg-dir", default=None, help="directory create_dataset_gt already wrote (for --only ppsg)")
    args = ap.parse_args()
    mods = ref_loader.load()
    if mods is None:
        sys.exit("reference checkout not found at %s" % ref_loader.REFERENCE_DIR)
    tools, pack, generate = mods
    import torch
    torch.manual_seed(0)
    want = lambda k: args.only is None or k in args.only  # noqa: E731
    if want("lbg2d"): make_lbg2d(tools)
    if want("lbg3d"): make_lbg3d(tools)
    if want("lb_legacy"): make_lb_legacy(tools)
    if want("macs2d"): make_macs2d(tools)
    if want("macs3d"): make_macs3d(tools)
    if args.only and "ppsg" in args.only: make_ppsg(tools, pack, args.ppsg_dir)   # slow: only on request
    if want("ppsg3d"): make_ppsg3d(pack, generate)
    if want("ppsg2d"): make_ppsg2d(generate)
    if want("ppsg2d_arm"): make_ppsg2d_arm(pack, generate)
    if want("stable3d"): make_stable3d(tools)
    if want("stable3d_wide"): make_stable3d_wide(tools)
    if want("kat"): make_kat(tools)
    if want("rolling"): make_rolling(tools, generate)
    if want("rolling_big"): make_rolling(tools, generate, ROLLING_BIG_SHAPES, "rolling_big.npz", seed=78)
    if want("render"): make_render(pack)
    if want("dataset_wide"): make_dataset_wide(pack)
    if want("data"):
        with tempfile.TemporaryDirectory() as tmp:
            statics, dynamics, tours = {}, {}, {}
            for D, count in ((2, 256), (3, 64)):
                statics[D], dynamics[D] = make_dataset(pack, D, tmp, count)
                make_masks(pack, D, statics[D], dynamics[D])
                tours[D] = make_episode(tools, pack, D, statics[D], dynamics[D])
            make_reward_tour(pack, tours, statics)


if __name__ == "__main__":
    main()
