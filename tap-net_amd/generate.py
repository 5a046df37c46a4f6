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
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().tap_pack_blocks(c, C.byref(desc), B, n, _lib.ptr(blocks), _lib.ptr(rew),
                                              _lib.ptr(pos), _lib.ptr(st), None, _lib.stream_of(dev)), c)
    return pos, st.bool(), rew


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
