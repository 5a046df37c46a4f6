"""Device-side RAND instance generation -- the counterpart of the reference's
``generate.generate_blocks`` (generate.py:773-971, ``container_width >= 0`` branch) and of what
``pack.create_dataset`` + ``PACKDataset`` make of it, without the text files in between.

Per instance the reference draws n block sizes, packs them into an *initial* container with
LB_GREEDY in hard mode and re-draws until every block is placed and stable
(generate.py:893-910), then reads the precedence relations off the packed container
(generate.py:913-914) and stores all D! rotations (generate.py:935-971).  Here a whole batch is
drawn, packed (one launch, ``tap_pack_blocks``), filtered and topped up until B instances are
accepted; one more launch (``tap_precedence``) writes ``static`` / ``dynamic`` in PACKDataset's
layout.  The random stream is torch's, not numpy's, so instances are equal in distribution, not
sample-for-sample; given the same block sizes the outputs are bit-identical to the reference's
(tests/test_gpu_parity.py::test_instances_from_reference_blocks).
"""
import ctypes as C

import torch

from . import _lib

RAND_PROB_4 = (0.15, 0.35, 0.35, 0.15)              # generate.py:880-882
RAND_PROB_5 = (0.08, 0.26, 0.32, 0.26, 0.08)        # generate.py:883-884


def size_distribution(size_range):
    """Probabilities over range(min_size, max_size) as generate.py:795, 880-890 picks them."""
    lo, hi = int(size_range[0]), int(size_range[1])
    k = hi - lo
    if k == 4:
        p = torch.tensor(RAND_PROB_4, dtype=torch.float64)
    elif k == 5:
        p = torch.tensor(RAND_PROB_5, dtype=torch.float64)
    else:
        mu, sigma = 0.5, 0.16
        x = torch.linspace(mu - 3 * sigma, mu + 3 * sigma, k, dtype=torch.float64)
        p = torch.exp(-(x - mu) ** 2 / (2 * sigma ** 2))
        p = p / p.sum()
    return torch.arange(lo, hi), p


def initial_container(block_dim, width, height):
    return [int(width), int(height)] if block_dim == 2 else [int(width), int(width), int(height)]


def pack_blocks(blocks, container_size, reward_type='C+P+S-lb-hard'):
    """tools.calc_positions_lb_greedy for a batch of explicit block lists (B, n, D) int32 on the
    device -> (positions (B,n,D) int32, stable (B,n) bool, neg_ratio (B,) float32)."""
    blocks = blocks.to(torch.int32).contiguous()
    B, n, D = blocks.shape
    dev = _lib.resolve_device(blocks.device)
    desc = _lib.make_desc(B, container_size, n, reward_type, 'full', 'LB_GREEDY')
    pos = torch.empty(B, n, D, dtype=torch.int32, device=dev)
    st = torch.empty(B, n, dtype=torch.uint8, device=dev)
    rew = torch.empty(B, dtype=torch.float32, device=dev)
    c = _lib.ctx(dev)
    try:
        # one launch for every container size (generate.py:908 accepts any --initial_container_width): lane-per-cell
        # groups up to 64 cells, one wavefront per container above (big.hip: k_big_wave_episode)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().tap_pack_blocks(c, C.byref(desc), B, n, _lib.ptr(blocks), _lib.ptr(rew),
                                                  _lib.ptr(pos), _lib.ptr(st), None, _lib.stream_of(dev)), c)
        return pos, st.bool(), rew
    except _lib.TapError as e:
        if e.status != _lib.TAP_E_UNSUPPORTED or not (desc.W * desc.L > 64 or (desc.D == 3 and (desc.W > 8 or desc.L > 8))):
            raise
    return _pack_blocks_stepped(blocks, container_size, reward_type, dev)


def _pack_blocks_stepped(blocks, container_size, reward_type, dev):
    """pack_blocks without a whole-episode kernel (TAP_NO_WAVE_KERNELS): the same episode as n placement launches on a
    state blob; a block with a side < 1 is not in the list (tap_pack_blocks' convention); NaN reward where a container
    raised an error bit."""
    from .env import BatchedContainer
    B, n, D = blocks.shape
    env = BatchedContainer(B, container_size, n, reward_type, 'full', packing_strategy='LB_GREEDY', device=dev)
    active = (blocks >= 1).all(dim=2)                                               # (B, n): is list entry t a block
    for t in range(n):
        env.add_new_blocks(blocks[:, t].contiguous(), active=active[:, t].contiguous(), want_feature=False)
    cps = env.calc_CPS()
    score = torch.where(env.counters[:, 3] > 0, (cps[:, 0] + cps[:, 1]) + cps[:, 2], torch.zeros_like(cps[:, 0]))
    rew = torch.where(env.errors != 0, torch.full_like(score, float('nan')), -score).to(torch.float32)
    # the container files its placements by placement count; tap_pack_blocks writes list index t (zeros for the entries
    # that are not blocks): the k-th active entry of a list takes slot k - 1
    slot = (active.long().cumsum(dim=1) - 1).clamp_(min=0)
    pos = torch.gather(env.positions, 1, slot.unsqueeze(2).expand(B, n, D))
    pos = torch.where(active.unsqueeze(2), pos, torch.zeros_like(pos))
    st = torch.gather(env.stable.to(torch.uint8), 1, slot).bool() & active
    return pos, st, rew


def precedence_tensors(blocks, positions, container_size, arm_size=1):
    """generate.calc_dependent + rotations + PACKDataset layout -> (static (B,1+D,nR),
    dynamic (B,3n,nR)) float32 for fully packed initial containers."""
    blocks = blocks.to(torch.int32).contiguous()
    positions = positions.to(device=blocks.device, dtype=torch.int32).contiguous()
    B, n, D = blocks.shape
    R = 2 if D == 2 else 6
    dev = _lib.resolve_device(blocks.device)
    static = torch.empty(B, 1 + D, n * R, dtype=torch.float32, device=dev)
    dynamic = torch.empty(B, 3 * n, n * R, dtype=torch.float32, device=dev)
    cs = (C.c_int32 * D)(*[int(v) for v in container_size])
    c = _lib.ctx(dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().tap_precedence(c, B, D, n, cs, int(arm_size), _lib.ptr(blocks),
                                             _lib.ptr(positions), _lib.ptr(static), _lib.ptr(dynamic),
                                             _lib.stream_of(dev)), c)
    return static, dynamic


def generate_instances(batch_size, blocks_num, block_dim, initial_container_width=7,
                       initial_container_height=50, arm_size=1, size_range=(1, 5), seed=None,
                       device='cuda', oversample=1.15, return_aux=False):
    """B accepted RAND instances -> (static, dynamic) as PACKDataset would hold them
    (plus blocks, positions with ``return_aux``)."""
    dev = _lib.resolve_device(device)
    gen = torch.Generator(device=dev)
    if seed is not None:
        gen.manual_seed(int(seed))
    values, prob = size_distribution(size_range)
    values, prob = values.to(dev), prob.to(device=dev, dtype=torch.float32)
    cs = initial_container(block_dim, initial_container_width, initial_container_height)
    got_blocks, got_pos, have = [], [], 0
    while have < batch_size:
        m = max(256, int((batch_size - have) * oversample) + 64)
        idx = torch.multinomial(prob, m * blocks_num * block_dim, replacement=True, generator=gen)
        blocks = values[idx].view(m, blocks_num, block_dim).to(torch.int32)          # generate.py:896
        pos, stable, rew = pack_blocks(blocks, cs)                                   # generate.py:908
        # generate.py:909-910; a packing that reaches above the initial container (NaN reward: the
        # reference would raise or clip its voxel grid before calc_dependent) is not an instance either
        ok = stable.all(dim=1) & ~torch.isnan(rew)
        got_blocks.append(blocks[ok]); got_pos.append(pos[ok])
        have += int(ok.sum().item())
    blocks = torch.cat(got_blocks)[:batch_size].contiguous()
    positions = torch.cat(got_pos)[:batch_size].contiguous()
    static, dynamic = precedence_tensors(blocks, positions, cs, arm_size)            # generate.py:913-971
    if return_aux:
        return static, dynamic, blocks, positions
    return static, dynamic


# ---- perfect-packing ("PPSG") instances: generate.generate_blocks_with_GT (generate.py:17-230) ---------------

def height_distribution(block_dim, blocks_num, size_range, initial_container_width, target_container_width,
                        samples=10000, seed=12345, device='cuda'):
    """generate.generate_height_prob (generate.py:977-1041): the distribution of perfect-packing heights,
    int(volume of a 10-block RAND instance / container bottom * blocks_num / 10) over ``samples`` accepted RAND
    instances -- the reference reads them from its 10 000-sample validation file, here they come from the
    device generator.  -> (prob, key) tensors, keys ascending."""
    _, _, blocks, _ = generate_instances(samples, 10, block_dim, initial_container_width, 50, 1, size_range,
                                         seed=seed, device=device, return_aux=True)
    bottom = target_container_width if block_dim == 2 else target_container_width * target_container_width
    vol = blocks.to(torch.int64).prod(dim=2).sum(dim=1)
    key = (vol.to(torch.float64) / bottom * (blocks_num / 10)).to(torch.int64)       # generate.py:1010
    keys, counts = torch.unique(key, return_counts=True)
    # A box W x W x H only splits into blocks_num blocks with every side < max_size if
    # ceil(W/(max_size-1))^(D-1) * ceil(H/(max_size-1)) <= blocks_num (pick points more than max_size-1 apart:
    # no two share a block).  For other heights generate_blocks_with_GT's acceptance loop (generate.py:66-73)
    # never ends -- the reference hangs there; those keys are dropped and the rest re-normalised.
    lim = int(size_range[1]) - 1
    per_layer = (-(-target_container_width // lim)) ** (block_dim - 1)
    ok = (per_layer * ((keys + lim - 1) // lim) <= blocks_num) & (keys >= 1)
    keys, counts = keys[ok], counts[ok]
    return counts.to(torch.float64) / counts.sum(), keys


def gauss_split_table(max_len, size_range=(1, 5), device='cuda'):
    """The table BPP_Generator_2D_easy's Gaussian split draws from (generate.py:463-469), built with numpy as the
    reference builds it -- linspace, exp, the normalisation, and the cumulative sum np.random.choice makes of the
    weights -- one row per side length L >= 2*max_size - 1.  -> float64 (max_len + 1, max_len) on ``device``."""
    import numpy as np
    mn, mx = int(size_range[0]), int(size_range[1])
    tab = np.zeros((max_len + 1, max(1, max_len)), np.float64)
    mu, sigma = 0.5, 0.16
    for side in range(2 * mx - 1, max_len + 1):
        m = side - 2 * mn
        if m < 1:
            continue
        prob_x = np.linspace(mu - 3 * sigma, mu + 3 * sigma, m)
        prob = np.exp(-(prob_x - mu) ** 2 / (2 * sigma ** 2)) / (np.sqrt(2 * np.pi) * sigma)
        prob = prob / np.sum(prob)
        cdf = prob.cumsum()
        tab[side, :m] = cdf / cdf[-1]
    return torch.as_tensor(tab, device=device)


def generate_ppsg_instances_2d(batch_size, blocks_num, initial_container_width=7, initial_container_height=50,
                               target_container_width=7, size_range=(1, 5), seed=12345, start=0, heights=None,
                               device='cuda', max_generations=4000, return_stats=False, input_type='bot',
                               mean_block_area=(4.9, 8.4), arm_size=1):
    """B perfect-packing instances (2D) as generate.generate_blocks_with_GT builds them for block_dim 2
    (generate.py:57-161 with BPP_Generator_2D_easy, :392-484): a guillotine-cut perfect packing of a
    target_width x H box, a random take-apart order with random rotations, packed in that order into the initial
    container with hard LB_GREEDY, accepted when all blocks are stable and can be taken out again in reverse;
    up to 20 orders per perfect packing, then a new one.  -> (blocks, positions) (B, n, 2) int32, layout order.
    Instances are keyed by global index ``start + i``.  ``heights`` (B,) overrides the heights drawn from
    height_distribution().

    The acceptance test is selective and its rate depends on the height (measured with these kernels, 20 blocks,
    W = 7, 'bot': one layout in 2.4e-5 .. 8e-5 accepted at H = 10 .. 12, 3e-4 .. 5e-4 at 14 .. 16, 2e-3 .. 5e-3 at
    20 .. 24; from H = 27 the cut generator itself accepts fewer than 1e-3 of its draws) -- the reference's own loop
    simply runs that long.  When the heights are drawn here, the distribution is therefore restricted to those
    with ``mean_block_area[0] <= W*H/n <= mean_block_area[1]`` (H = 14 .. 24 at 20 blocks: 90 % of the reference's
    distribution) and re-normalised; pass ``mean_block_area=None`` for all of it.

    ``arm_size`` (generate.py:623-641, 2D only): columns the arm needs beside a block; it enters the take-apart test
    through the left / right relations (the take-apart ORDER of the perfect packing uses movement dependences only,
    generate.py:77, which do not depend on it)."""
    dev = _lib.resolve_device(device)
    n, B, W = int(blocks_num), int(batch_size), int(target_container_width)
    if n > 64:
        raise ValueError("blocks_num <= 64")
    ids = torch.arange(start, start + B, dtype=torch.int64, device=dev)
    if heights is None:
        prob, keys = height_distribution(2, n, size_range, initial_container_width, W, seed=seed, device=dev)
        if mean_block_area is not None:
            area = keys.to(torch.float64) * W / n
            keep = (area >= mean_block_area[0]) & (area <= mean_block_area[1])
            if not bool(keep.any()):
                raise ValueError("no height of the distribution has a mean block area in %r" % (mean_block_area,))
            prob, keys = prob[keep] / prob[keep].sum(), keys[keep]
        g = torch.Generator(device='cpu')
        rows = []
        for i in range(B):                               # keyed per instance id so shards agree
            g.manual_seed((int(seed) * 1000003 + start + i) % (2 ** 63))
            rows.append(torch.multinomial(prob.cpu().float(), 1, replacement=True, generator=g))
        heights = keys.cpu()[torch.cat(rows)]
    heights = heights.to(device=dev, dtype=torch.int32).contiguous().view(-1)
    if int(heights.numel()) != B:
        raise ValueError("heights must hold %d values" % B)
    lim = int(size_range[1]) - 1
    need = (-(-W // lim)) * ((heights + lim - 1) // lim)
    if bool(((need > n) | (heights < 1) | (W * heights < n)).any()):
        raise ValueError("a height cannot be cut into %d blocks with sides < %d" % (n, int(size_range[1])))
    gauss = gauss_split_table(int(max(W, int(heights.max().item()))), size_range, dev)
    cs = initial_container(2, initial_container_width, initial_container_height)
    c, L = _lib.ctx(dev), _lib.lib()
    out_blocks = torch.zeros(B, n, 2, dtype=torch.int32, device=dev)
    out_pos = torch.zeros(B, n, 2, dtype=torch.int32, device=dev)
    done = torch.zeros(B, dtype=torch.bool, device=dev)
    stats = dict(generations=0, layouts=0)
    from .rolling import RollingWindows
    for gen in range(max_generations):
        todo = (~done).nonzero().squeeze(1)
        m = int(todo.numel())
        if m == 0:
            break
        stats['generations'] = gen + 1
        tid, th = ids[todo].contiguous(), heights[todo].contiguous()
        gtb = torch.empty(m, n, 2, dtype=torch.int32, device=dev)
        gtp = torch.empty(m, n, 2, dtype=torch.int32, device=dev)
        att = torch.empty(m, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.tap_ppsg_gt2d(c, m, n, W, _lib.ptr(th), int(size_range[0]), int(size_range[1]), _lib.ptr(gauss),
                                       int(gauss.shape[1]), int(gauss.shape[0]), int(seed), _lib.ptr(tid), 0, gen, 1 << 20,
                                       _lib.ptr(gtb), _lib.ptr(gtp), _lib.ptr(att), _lib.stream_of(dev)), c)
        if bool((att < 0).any()):
            raise _lib.TapError(_lib.TAP_E_INVALID, "a perfect packing was not found within the attempt cap")
        alive = torch.ones(m, dtype=torch.bool, device=dev)
        for trial in range(20):                                                     # generate.py:83-87
            sel = alive.nonzero().squeeze(1)
            k = int(sel.numel())
            if k == 0:
                break
            stats['layouts'] += k
            blocks = torch.empty(k, n, 2, dtype=torch.int32, device=dev)
            sb, sp, si = gtb[sel].contiguous(), gtp[sel].contiguous(), tid[sel].contiguous()   # named: see the 3D form
            with torch.cuda.device(dev):
                _lib.check(L.tap_ppsg_order2d(c, k, n, _lib.ptr(sb), _lib.ptr(sp), int(seed), _lib.ptr(si), 0, gen, trial,
                                              _lib.ptr(blocks), _lib.stream_of(dev)), c)
            pos, stable, rew = pack_blocks(blocks, cs)                               # generate.py:108
            rw = RollingWindows(blocks, pos, cs, child_graph_size=1, arm_size=int(arm_size))   # the relations (:111-112)
            ok = torch.empty(k, dtype=torch.uint8, device=dev)
            st8 = stable.to(torch.uint8).contiguous()
            with torch.cuda.device(dev):
                _lib.check(L.tap_ppsg_check(c, k, n, 1 if input_type == 'simple' else 0, _lib.ptr(rw.rel), _lib.ptr(st8),
                                            _lib.ptr(ok), _lib.stream_of(dev)), c)
            good = ok.bool() & ~torch.isnan(rew)
            hit = sel[good]
            out_blocks[todo[hit]] = blocks[good]
            out_pos[todo[hit]] = pos[good]
            done[todo[hit]] = True
            alive[hit] = False
    if not bool(done.all()):
        raise _lib.TapError(_lib.TAP_E_INVALID, "%d PPSG instances not found in %d generations" % (int((~done).sum()), max_generations))
    if return_stats:
        return out_blocks, out_pos, stats
    return out_blocks, out_pos


def generate_ppsg_instances(batch_size, blocks_num, initial_container_width=7, initial_container_height=50,
                            target_container_width=5, size_range=(1, 5), seed=12345, start=0, slab_blocks=10,
                            heights=None, device='cuda', max_generations=50, return_stats=False, input_type='bot',
                            arm_size=1):
    """B perfect-packing instances (3D) as generate.generate_blocks_with_GT builds them (generate.py:57-161):
    a guillotine-cut perfect packing of a target_width^2 x H box (BPP_Generator_3D + its acceptance test), a
    random take-apart order with random rotations, the blocks packed in that order into the initial container
    with hard LB_GREEDY, accepted when all are stable and can be taken out again in reverse; up to 20 orders
    per perfect packing, then a new one.  -> (blocks, positions) (B, n, 3) int32 on ``device``, layout order.

    blocks_num > slab_blocks: the perfect packing is blocks_num / slab_blocks stacked slabs, each cut by the
    reference's generator (its single rejection loop accepts < 2e-8 of its draws at 50 blocks and cannot
    produce such an instance).  Instances are keyed by global index ``start + i``: any sharding sees the same
    ones.  ``heights`` (B, S) overrides the per-slab heights drawn from height_distribution().

    ``input_type`` selects the take-apart test as the reference does (generate.py:135-141): 'simple' asks only
    that nothing rests on a block when its turn comes; every other type ('bot') also asks for one free side per
    horizontal axis.  Measured with the pinned restatement: the 'bot' form accepts about 1 % of the stable
    layouts at 10 blocks and none of 200 at 20 or 50 blocks, so instances above ~10 blocks need 'simple'."""
    import ctypes as C
    dev = _lib.resolve_device(device)
    n, B = int(blocks_num), int(batch_size)
    ns = min(int(slab_blocks), n)
    if n % ns:
        raise ValueError("blocks_num must be a multiple of slab_blocks")
    S, W = n // ns, int(target_container_width)
    ids = torch.arange(start, start + B, dtype=torch.int64, device=dev)
    if heights is None:
        prob, keys = height_distribution(3, ns, size_range, initial_container_width, W, seed=seed, device=dev)
        g = torch.Generator(device='cpu')
        rows = []
        for i in range(B):                               # keyed per instance id so shards agree
            g.manual_seed((int(seed) * 1000003 + start + i) % (2 ** 63))
            rows.append(torch.multinomial(prob.cpu().float(), S, replacement=True, generator=g))
        heights = keys.cpu()[torch.stack(rows)]
    heights = heights.to(device=dev, dtype=torch.int32).contiguous()
    if tuple(heights.shape) != (B, S):
        raise ValueError("heights must be (%d, %d)" % (B, S))
    # a W x W x h box needs ceil(W/lim)^2 * ceil(h/lim) blocks to keep every side below max_size: for other
    # heights the acceptance loop of generate.py:66-73 never ends (and the kernel would spin to its attempt cap)
    lim = int(size_range[1]) - 1
    need = (-(-W // lim)) ** 2 * ((heights + lim - 1) // lim)
    if bool(((need > ns) | (heights < 1)).any()):
        raise ValueError("a slab height cannot be cut into %d blocks with sides < %d" % (ns, int(size_range[1])))
    cs = initial_container(3, initial_container_width, initial_container_height)
    c, L = _lib.ctx(dev), _lib.lib()
    out_blocks = torch.zeros(B, n, 3, dtype=torch.int32, device=dev)
    out_pos = torch.zeros(B, n, 3, dtype=torch.int32, device=dev)
    done = torch.zeros(B, dtype=torch.bool, device=dev)
    stats = dict(generations=0, layouts=0)
    from .rolling import RollingWindows
    for gen in range(max_generations):
        todo = (~done).nonzero().squeeze(1)
        m = int(todo.numel())
        if m == 0:
            break
        stats['generations'] = gen + 1
        tid, th = ids[todo].contiguous(), heights[todo].contiguous()
        gtb = torch.empty(m, n, 3, dtype=torch.int32, device=dev)
        gtp = torch.empty(m, n, 3, dtype=torch.int32, device=dev)
        att = torch.empty(m, S, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.tap_ppsg_gt(c, m, S, ns, W, _lib.ptr(th), int(size_range[0]), int(size_range[1]), int(seed),
                                     _lib.ptr(tid), 0, gen, 1 << 20, _lib.ptr(gtb), _lib.ptr(gtp), _lib.ptr(att),
                                     _lib.stream_of(dev)), c)
        if bool((att < 0).any()):
            raise _lib.TapError(_lib.TAP_E_INVALID, "a perfect packing was not found within the attempt cap")
        alive = torch.ones(m, dtype=torch.bool, device=dev)
        for trial in range(20):                                                     # generate.py:83-87
            sel = alive.nonzero().squeeze(1)
            k = int(sel.numel())
            if k == 0:
                break
            stats['layouts'] += k
            blocks = torch.empty(k, n, 3, dtype=torch.int32, device=dev)
            # named temporaries: an unnamed one is freed as soon as its pointer is taken and the next one would
            # be handed the same memory
            sb, sp, si = gtb[sel].contiguous(), gtp[sel].contiguous(), tid[sel].contiguous()
            with torch.cuda.device(dev):
                _lib.check(L.tap_ppsg_order(c, k, n, _lib.ptr(sb), _lib.ptr(sp), int(seed), _lib.ptr(si), 0, gen, trial,
                                            _lib.ptr(blocks), _lib.stream_of(dev)), c)
            pos, stable, rew = pack_blocks(blocks, cs)                               # generate.py:108
            # (arm_size is read by calc_dependent's 2D rule only, generate.py:623-641; passed on for symmetry)
            rw = RollingWindows(blocks, pos, cs, child_graph_size=1, arm_size=int(arm_size))   # the five relations (:111-112)
            ok = torch.empty(k, dtype=torch.uint8, device=dev)
            st8 = stable.to(torch.uint8).contiguous()
            with torch.cuda.device(dev):
                _lib.check(L.tap_ppsg_check(c, k, n, 1 if input_type == 'simple' else 0, _lib.ptr(rw.rel), _lib.ptr(st8),
                                            _lib.ptr(ok), _lib.stream_of(dev)), c)
            good = ok.bool() & ~torch.isnan(rew)
            hit = sel[good]
            out_blocks[todo[hit]] = blocks[good]
            out_pos[todo[hit]] = pos[good]
            done[todo[hit]] = True
            alive[hit] = False
    if not bool(done.all()):
        raise _lib.TapError(_lib.TAP_E_INVALID, "%d PPSG instances not found in %d generations" % (int((~done).sum()), max_generations))
    if return_stats:
        return out_blocks, out_pos, stats
    return out_blocks, out_pos


def generate_mix_instances(batch_size, blocks_num, block_dim=3, initial_container_width=7, initial_container_height=50,
                           seed=12345, start=0, device='cuda', target_container_width=5, size_range=(1, 5),
                           input_type=None):
    """The MIX series as PACKDataset mixes its two files (pack.py:67-97): the first half of the batch from the
    perfect-packing generator, the second half RAND.  -> (blocks, positions) (B, n, D) int32.
    ``input_type``: the PPSG half's take-apart test (see generate_ppsg_instances); default 'bot' up to 10
    blocks, 'simple' above (where 'bot' accepts nothing)."""
    if input_type is None:
        input_type = 'bot' if int(blocks_num) <= 10 else 'simple'
    half = int(batch_size) // 2
    if block_dim == 2:
        pb, pp = generate_ppsg_instances_2d(half, blocks_num, initial_container_width, initial_container_height,
                                            target_container_width, size_range, seed=seed, start=start, device=device,
                                            input_type=input_type)
    else:
        pb, pp = generate_ppsg_instances(half, blocks_num, initial_container_width, initial_container_height,
                                         target_container_width, size_range, seed=seed, start=start, device=device,
                                         input_type=input_type)
    _, _, rb, rp = generate_instances(int(batch_size) - half, blocks_num, block_dim, initial_container_width,
                                      initial_container_height, 1, size_range, seed=seed, device=device, return_aux=True)
    return torch.cat([pb, rb]).contiguous(), torch.cat([pp, rp]).contiguous()


def __getattr__(name):
    # generate.InitialContainer (generate.py:1589): the per-instance facade lives with the batched windows (rolling.py,
    # which imports this module's callers); resolved on first use
    if name == 'InitialContainer':
        from .rolling import InitialContainer
        return InitialContainer
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
