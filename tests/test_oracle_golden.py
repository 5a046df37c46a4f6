"""Pin the CPU oracle against golden vectors produced by the reference itself (CPU only)."""
import numpy as np
import pytest

import golden_util as G
import oracle_lib as O


def _check_trace(meta, tr):
    desc = O.make_desc(meta["cs"], meta["n"], meta["reward"], meta["feat"], meta["strategy"])
    out = O.run_episodes(desc, tr["blocks"].astype(np.int32))
    assert out["nerr"] == 0, meta
    B, n = tr["blocks"].shape[:2]
    assert np.array_equal(out["positions"], tr["positions"]), meta
    assert np.array_equal(out["stable"], tr["stable"]), meta
    assert np.array_equal(out["features"], tr["features"].reshape(B, n, -1)), meta
    assert np.array_equal(out["heightmaps"], tr["heightmaps"]), meta
    assert np.array_equal(out["counters"][:, 0], tr["valid"][:, -1]), meta
    assert np.array_equal(out["counters"][:, 1], tr["empty"][:, -1]), meta
    # fp64 ratios bit-for-bit (NaN where the reference gives NaN)
    assert np.array_equal(out["ratio"].view(np.int64), tr["ratio"].view(np.int64)) or \
        np.array_equal(np.nan_to_num(out["ratio"], nan=-7.0), np.nan_to_num(tr["ratio"], nan=-7.0)), meta
    assert np.array_equal(np.nan_to_num(out["cps"], nan=-7.0), np.nan_to_num(tr["cps"], nan=-7.0)), meta


@pytest.mark.parametrize("fixture", ["lbg2d.npz", "lbg3d.npz", "macs2d.npz", "macs3d.npz", "lb_legacy.npz"])
def test_container_traces(fixture):
    ncases = 0
    for meta, tr in G.cases(fixture):
        _check_trace(meta, tr)
        ncases += 1
    assert ncases >= 8


def test_ppsg_instances_macs_traces():
    """The reference's own PPSG instances (perfect-packing generator) through MACS."""
    n = 0
    for meta, tr in G.cases("ppsg_2d.npz"):
        _check_trace(meta, tr)
        n += 1
    assert n == 2


def test_per_step_counters_2d():
    """valid/empty after every step, via the step-wise Env API."""
    for meta, tr in G.cases("lbg2d.npz"):
        env = O.Env(meta["cs"], meta["n"], meta["reward"], meta["feat"], meta["strategy"])
        for b in range(min(8, tr["blocks"].shape[0])):
            env.clear()
            for t in range(meta["n"]):
                rc, _ = env.add_new_block(tr["blocks"][b, t])
                assert rc == 0
                assert env.valid_size == tr["valid"][b, t] and env.empty_size == tr["empty"][b, t]


def test_is_stable_3d_exhaustive():
    z = G.load("stable3d.npz")
    bits = np.unpackbits(z["bits"])[: int(z["count"])]
    checked = 0
    for (bx, by), off in zip(z["shapes"], z["offsets"]):
        bx, by = int(bx), int(by)
        cells = bx * by
        key = "m_%d_%d" % (bx, by)
        masks = z[key] if key in z.files else np.arange(1 << cells, dtype=np.int64)
        for mi, m in enumerate(masks):
            lay = np.array([(int(m) >> k) & 1 for k in range(cells)], dtype=np.uint8)
            assert O.is_stable_3d_mask(bx, by, lay) == bool(bits[off + mi]), (bx, by, int(m))
            checked += 1
    assert checked == int(z["count"]) and checked > 100000


def test_known_answers():
    k = G.load("kat.npz")
    rc, pos, st, ratio, scores = O.calc_positions_lb_greedy(k["k2_blocks"], [5, 50], "C+P+S-lb-soft")
    assert rc == 0 and np.array_equal(pos, k["k2_pos"]) and np.array_equal(st, k["k2_stable"])
    assert ratio == float(k["k2_ratio"]) and np.array_equal(scores, k["k2_scores"])
    # SURVEY appendix G, human-checkable
    assert pos.tolist() == [[0, 0], [3, 0], [4, 0], [0, 2], [0, 6]] and st.tolist() == [1, 1, 1, 1, 0]
    rc, pos, st, ratio, scores = O.calc_positions_lb_greedy(k["k3_blocks"], [5, 5, 50], "C+P+S-lb-soft")
    assert rc == 0 and np.array_equal(pos, k["k3_pos"]) and np.array_equal(st, k["k3_stable"])
    assert ratio == float(k["k3_ratio"]) == 1.9197701149425286 and scores.tolist() == [46, 100, 12, 4, 4]
    for tag, reward in (("soft", "C+P+S-lb-soft"), ("hard", "C+P+S-lb-hard"), ("cp", "C+P-lb-hard")):
        rc, pos, st, ratio, scores = O.calc_positions_lb_greedy(k["k7_blocks"], [7, 50], reward)
        assert rc == 0 and np.array_equal(pos, k["k7_%s_pos" % tag])
        assert np.array_equal(st, k["k7_%s_stable" % tag]) and ratio == float(k["k7_%s_ratio" % tag])
        assert np.array_equal(scores, k["k7_%s_scores" % tag])
    # a block wider than the container: no-op, counters untouched, step still counted, ratio NaN
    env = O.Env([3, 20], 2, "C+P+S-lb-soft", "diff")
    rc, f = env.add_new_block([4, 1])
    assert rc == 0 and np.array_equal(f, k["fail_feature"]) and env.valid_size == int(k["fail_valid"]) == 0
    assert np.isnan(env.calc_ratio()) and np.isnan(float(k["fail_ratio"]))


def test_height_overflow_flag():
    env = O.Env([2, 6], 4, "C+P+S-lb-soft", "diff")
    for _ in range(3):
        rc, _ = env.add_new_block([2, 2])
        assert rc == 0
    rc, _ = env.add_new_block([2, 2])          # 6 + 2 > H
    assert rc == -2 and env.error == -2


@pytest.mark.parametrize("D", [2, 3])
def test_masks_trace(D):
    ds = G.load("dataset_%dd.npz" % D)
    mk = G.load("masks_%dd.npz" % D)
    static = ds["static"].astype(np.float32)
    dynamic = ds["dynamic"].astype(np.float32)
    B, rows, nR = dynamic.shape
    n = rows // 3
    R = nR // n
    cur = O.initial_mask(dynamic, n)
    assert np.array_equal(cur, mk["initial_mask"].astype(np.float32))
    mask = np.ones((B, nR), np.float32)
    dyn = dynamic
    for t in range(mk["ptr"].shape[0]):
        ptr = mk["ptr"][t].astype(np.int64)
        dyn = O.update_dynamic(dyn, static, ptr, n, 3)
        cur, mask = O.update_mask(mask, dyn, ptr, n, R)
        assert np.array_equal(dyn, G.unpack_dynamic(mk["dynamic_bits"][t], dynamic.shape)), t
        assert np.array_equal(cur, mk["current_mask"][t].astype(np.float32)), t
        assert np.array_equal(mask, mk["mask"][t].astype(np.float32)), t
    assert not mask.any()


@pytest.mark.parametrize("D", [2, 3])
def test_episode_trace(D):
    """Replay the reference actor's greedy tour: features, masks, dynamic and scores must match."""
    ds = G.load("dataset_%dd.npz" % D)
    ep = G.load("episode_%dd.npz" % D)
    static = ds["static"].astype(np.float32)
    dynamic = ds["dynamic"].astype(np.float32)
    B, rows, nR = dynamic.shape
    n = rows // 3
    R = nR // n
    tour = ep["tour_idx"].astype(np.int64)
    cs = [5, 50] if D == 2 else [5, 5, 50]
    envs = [O.Env(cs, n, "C+P+S-lb-soft", "diff") for _ in range(B)]
    mask = np.ones((B, nR), np.float32)
    dyn = dynamic
    for t in range(tour.shape[1]):
        ptr = tour[:, t]
        dyn = O.update_dynamic(dyn, static, ptr, n, 3)
        cur, mask = O.update_mask(mask, dyn, ptr, n, R)
        assert np.array_equal(dyn, G.unpack_dynamic(ep["dynamic_bits"][t], dynamic.shape))
        assert np.array_equal(cur, ep["current_mask"][t].astype(np.float32))
        assert np.array_equal(mask, ep["mask"][t].astype(np.float32))
        blocks = static[np.arange(B), 1:, ptr]                    # model.py:404-412
        for b in range(B):
            rc, f = envs[b].add_new_block(blocks[b])
            assert rc == 0
            assert np.array_equal(f.reshape(-1), ep["features"][t, b])
    scores = np.array([e.calc_ratio() for e in envs]).astype(np.float32)   # model.py:499,510
    assert np.array_equal(-scores, ep["neg_scores"])


def test_reward_tour():
    rt = G.load("reward_tour.npz")
    for D in (2, 3):
        ds = G.load("dataset_%dd.npz" % D)
        nerr, r = O.reward(ds["static"].astype(np.float32), rt["tour_%dd" % D].astype(np.int64),
                           "C+P+S-lb-soft", 5, 50)
        assert nerr == 0 and np.array_equal(r, rt["reward_%dd" % D])


def test_render_metric_files():
    """pack.render's eight files (pack.py:967-977) as the reference wrote them: the oracle's whole-episode figures
    (calc_positions_lb_greedy / calc_positions_mcs, two-container averaging) re-written with np.savetxt are the same
    bytes."""
    import io
    ncases = 0
    for meta, static, tour, files in G.render_cases():
        ratio, scores, errs = O.render_scores(static, tour, meta["reward"], meta["input_type"], True, meta["W"], meta["H"],
                                              meta["strategy"], initial_container_height=meta["H0"])
        assert not errs.any(), meta
        cols = dict(ratio=ratio, valid_size=scores[:, 0], box_size=scores[:, 1], empty_size=scores[:, 2],
                    stable_num=scores[:, 3], packing_height=scores[:, 4], time=np.array([meta["valid_time"]]), ids=tour)
        for suf, want in files.items():
            buf = io.BytesIO()
            np.savetxt(buf, cols[suf])
            assert buf.getvalue() == want, (meta, suf)
        ncases += 1
    assert ncases == 16


@pytest.mark.parametrize("D", [2, 3])
def test_instance_generation_from_reference_datasets(D):
    """generate.generate_blocks + calc_dependent + PACKDataset layout: re-derive positions, static
    and dynamic from the block sizes of instances the reference generated itself."""
    ds = G.load("dataset_%dd.npz" % D)
    n = 10
    R = 2 if D == 2 else 6
    N = ds["static"].shape[0]
    blocks_txt = ds["txt_blocks"].reshape(N, R, D, n)              # R lines per sample, dimension-major
    pos_txt = ds["txt_pos"].reshape(N, D, n)
    init = [7, 50] if D == 2 else [7, 7, 50]
    for b in range(N):
        blocks = blocks_txt[b, 0].T                                 # rotation 0 = the sampled sizes
        rc, pos, st, dyn = O.instance_from_blocks(blocks, init, arm_size=1)
        assert rc == 1, b                                           # the reference accepted it
        assert np.array_equal(pos, pos_txt[b].T), b
        assert np.array_equal(st, ds["static"][b].astype(np.float32)), b
        assert np.array_equal(dyn, ds["dynamic"][b].astype(np.float32)), b


@pytest.mark.parametrize("fixture", ["rolling.npz", "rolling_big.npz"])
def test_rolling_windows(fixture):
    """generate.InitialContainer window traces (the outer loop of rolling.py); rolling_big.npz: instances of 70 .. 130
    blocks, where node ids above 63 meet the CPython-set iteration order."""
    import ast
    z = G.load(fixture)
    for i, m in enumerate(z["cases"]):
        meta = ast.literal_eval(str(m))
        tag = "r%d_" % i
        ro = O.Rolling(z[tag + "blocks"], z[tag + "pos"], meta["init"], meta["child"])
        T = z[tag + "static"].shape[0]
        for t in range(T):
            rc, st, dy, nodes = ro.convert_to_input()
            assert rc == (1 if t == T - 1 else 0), (i, t, rc)
            assert np.array_equal(nodes, z[tag + "nodes"][t]), (i, t)
            assert np.array_equal(st, z[tag + "static"][t].astype(np.float32)), (i, t)
            assert np.array_equal(dy, z[tag + "dynamic"][t].astype(np.float32)), (i, t)
            if t < T - 1:
                ro.remove(int(z[tag + "ptr"][t]) % meta["child"])


def test_bpp_generator_3d_draw_for_draw():
    """generate.BPP_Generator_3D (generate.py:232-301): fed the MT19937 words numpy's RandomState(seed) hands
    out, the restatement must make the reference's cuts draw for draw (accepted packings or not)."""
    z = G.load("ppsg3d.npz")
    naccepted = 0
    for k, (n, gx, gy, gz, seed) in enumerate(z["bpp_cases"]):
        rng = O.Rng(words=O.numpy_mt_words(seed, 4096))
        rc, blocks, pos = O.bpp3d(rng, int(n), [int(gx), int(gy), int(gz)])
        assert rc >= 0 and not rng.exhausted
        assert np.array_equal(blocks, z["bpp%d_blocks" % k]), (k, n, seed)
        assert np.array_equal(pos, z["bpp%d_positions" % k]), (k, n, seed)
        want_ok = bool(((z["bpp%d_blocks" % k] >= 1) & (z["bpp%d_blocks" % k] < 5)).all())
        assert bool(rc) == want_ok
        naccepted += rc
    assert len(z["bpp_cases"]) > 250 and naccepted > 0


def test_generate_blocks_with_gt_draw_for_draw():
    """generate.generate_blocks_with_GT (generate.py:17-230), 3D: rejection loop over BPP_Generator_3D, random
    unpacking order, random rotations, hard LB_GREEDY layout, stability and take-apart tests -- same word
    stream in, the reference's instance out (blocks / positions in layout order, and through
    instance_from_blocks the PACKDataset tensors the reference's own reader makes of its return values)."""
    z = G.load("ppsg3d.npz")
    for k, (n, gx, gy, gz, seed) in enumerate(z["gt_cases"]):
        n = int(n)
        rng = O.Rng(words=O.numpy_mt_words(seed, 6_000_000))
        rc, blocks, pos, stats = O.generate_blocks_with_gt(rng, n, [int(gx), int(gy), int(gz)], [7, 7, 50])
        assert rc == 1 and not rng.exhausted, (k, rc)
        assert np.array_equal(blocks, z["gt%d_blocks" % k]), (k, stats)
        assert np.array_equal(pos, z["gt%d_positions" % k]), (k, stats)
        ok, pos2, st, dyn = O.instance_from_blocks(blocks, [7, 7, 50], 1)
        assert ok == 1 and np.array_equal(pos2, pos)
        assert np.array_equal(st, z["gt%d_static" % k].astype(np.float32))
        assert np.array_equal(dyn, z["gt%d_dynamic" % k].astype(np.float32))
    assert len(z["gt_cases"]) == 24


def test_bpp_generator_2d_easy_draw_for_draw():
    """generate.BPP_Generator_2D_easy (generate.py:392-484), the 2D generator generate_blocks_with_GT calls: fed
    numpy's MT19937 words the restatement makes the reference's cuts draw for draw -- the weighted block and axis
    choices, the axis rule that compares a size with a POSITION (:447), uniform and Gaussian split positions (the
    Gaussian table is built by numpy on both sides)."""
    z = G.load("ppsg2d.npz")
    naccepted = 0
    for k, (n, gx, gz, seed) in enumerate(z["bpp_cases"]):
        rng = O.Rng(words=O.numpy_mt_words(seed, 4096))
        rc, blocks, pos = O.bpp2d_easy(rng, int(n), [int(gx), int(gz)])
        assert rc >= 0 and not rng.exhausted
        assert np.array_equal(blocks, z["bpp%d_blocks" % k]), (k, n, seed)
        assert np.array_equal(pos, z["bpp%d_positions" % k]), (k, n, seed)
        want_ok = bool(((z["bpp%d_blocks" % k] >= 1) & (z["bpp%d_blocks" % k] < 5)).all())
        assert bool(rc) == want_ok
        naccepted += rc
    assert len(z["bpp_cases"]) == 270 and naccepted > 20


def test_generate_blocks_with_gt_2d_draw_for_draw():
    """generate.generate_blocks_with_GT for block_dim 2: same word stream in, the reference's instance out."""
    z = G.load("ppsg2d.npz")
    for k, (n, gx, gz, seed) in enumerate(z["gt_cases"]):
        n = int(n)
        rng = O.Rng(words=O.numpy_mt_words(seed, 4_000_000))
        rc, blocks, pos, stats = O.generate_blocks_with_gt_2d(rng, n, [int(gx), int(gz)], [7, 50])
        assert rc == 1 and not rng.exhausted, (k, rc)
        assert np.array_equal(blocks, z["gt%d_blocks" % k]), (k, stats)
        assert np.array_equal(pos, z["gt%d_positions" % k]), (k, stats)
        ok, pos2, st, dyn = O.instance_from_blocks(blocks, [7, 50], 1)
        assert ok == 1 and np.array_equal(pos2, pos)
        # the function's own deps_move (row = blocked block); PACKDataset's tensor holds it transposed
        assert np.array_equal(dyn[:n, :n].T.reshape(-1), z["gt%d_dep_move" % k].astype(np.float32))
    assert len(z["gt_cases"]) == 20


def test_arm_sizes_above_one_2d():
    """`--arm_size` above one column (generate.py:623-641; 2D only): whole generate_blocks_with_GT runs with arm_size 2 / 3
    draw for draw, and the PACKDataset tensors of generate_blocks_with_GT / generate_blocks instances with arm_size 2 .. 4
    (the left / right access rules over `arm_size` columns, the wall rule at x < arm_size) through instance_from_blocks."""
    z = G.load("ppsg2d_arm.npz")
    for k, (n, gx, gz, w0, arm, seed) in enumerate(z["arm_cases"]):
        n, init = int(n), [int(w0), 50]
        if k < 17:                                           # the last three need > 4 M words (10^4 rejected packings)
            rng = O.Rng(words=O.numpy_mt_words(int(seed), 4_000_000))
            rc, blocks, pos, stats = O.generate_blocks_with_gt_2d(rng, n, [int(gx), int(gz)], init, arm_size=int(arm))
            assert rc == 1 and not rng.exhausted, (k, rc)
            assert np.array_equal(blocks, z["arm%d_blocks" % k]) and np.array_equal(pos, z["arm%d_positions" % k]), (k, stats)
        ok, pos2, st, dyn = O.instance_from_blocks(z["arm%d_blocks" % k], init, int(arm))
        assert ok == 1 and np.array_equal(pos2, z["arm%d_positions" % k]), k
        assert np.array_equal(st, z["arm%d_static" % k].astype(np.float32)), k
        assert np.array_equal(dyn, z["arm%d_dynamic" % k].astype(np.float32)), k
    for k, (n, w0, h0, arm, seed) in enumerate(z["rnd_cases"]):
        ok, pos2, st, dyn = O.instance_from_blocks(z["rnd%d_blocks" % k], [int(w0), int(h0)], int(arm))
        assert ok == 1 and np.array_equal(pos2, z["rnd%d_positions" % k]), k
        assert np.array_equal(st, z["rnd%d_static" % k].astype(np.float32)), k
        assert np.array_equal(dyn, z["rnd%d_dynamic" % k].astype(np.float32)), k
    assert len(z["arm_cases"]) == 20 and len(z["rnd_cases"]) == 24
    # the wider arm is live in the fixture: some instance's tensors differ from the arm_size 1 ones
    differs = 0
    for k, (n, w0, h0, arm, seed) in enumerate(z["rnd_cases"]):
        _, _, _, dyn1 = O.instance_from_blocks(z["rnd%d_blocks" % k], [int(w0), int(h0)], 1)
        differs += int(not np.array_equal(dyn1, z["rnd%d_dynamic" % k].astype(np.float32)))
    assert differs >= 12
