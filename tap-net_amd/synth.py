"""Synthetic inputs for benchmarks and tests (host-side numpy; results are moved to the device).

Streams are keyed by *global* env index in chunks of ``CHUNK`` envs, so any contiguous sharding of
the batch across ranks (aligned to CHUNK) sees exactly the data the single-GPU run sees.
"""
import itertools

import numpy as np
import torch

CHUNK = 1024
RAND_SIZES = np.array([1, 2, 3, 4])
RAND_PROB = np.array([0.15, 0.35, 0.35, 0.15])   # generate.py:881-882 with size_range [1, 5)


def _chunk_rng(seed, chunk):
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence([int(seed), int(chunk)])))


def _per_chunk(seed, start, count, fn):
    """Concatenate fn(rng, m) over the chunks covering envs [start, start+count)."""
    out = []
    e = start
    while e < start + count:
        c = e // CHUNK
        lo, hi = c * CHUNK, (c + 1) * CHUNK
        full = fn(_chunk_rng(seed, c), CHUNK)
        out.append(full[e - lo: min(hi, start + count) - lo])
        e = hi
    return np.concatenate(out, axis=0)


def rand_blocks(batch, n, block_dim, seed=12345, start=0):
    """(batch, n, D) int32 block sizes with the RAND marginal (generate.py:795,881-882,896)."""
    return _per_chunk(seed, start, batch, lambda rng, m: rng.choice(
        RAND_SIZES, size=(m, n, block_dim), p=RAND_PROB).astype(np.int32))


def rand_instances(batch, n, block_dim, seed=12345, start=0, p_move=0.22, p_side=0.15, p_self=0.3):
    """Synthetic packing instances in PACKDataset layout ('bot', allow_rot=True):
    static (batch, 1+D, n*R) float32 and dynamic (batch, 3n, n*R) float32.

    Precedence is a random DAG: blocks get a random stacking order and a block may rest on
    (``move``-block) only blocks earlier in that order, so every instance can be emptied; side
    (``small`` / ``large``) blockers incl. wall self-dependencies (generate.py:623-625) are random
    and exist only for rotations that change the vertical axis (generate.py:941-960).
    """
    perms = list(itertools.permutations(range(block_dim)))
    R = len(perms)

    def make(rng, m):
        sizes = rng.choice(RAND_SIZES, size=(m, n, block_dim), p=RAND_PROB).astype(np.float32)
        static = np.zeros((m, 1 + block_dim, n * R), np.float32)
        dynamic = np.zeros((m, 3 * n, n * R), np.float32)
        order = np.argsort(rng.random((m, n)), axis=1)              # order[b, k] = k-th lowest block
        rank = np.argsort(order, axis=1)                            # rank[b, i] = height rank of block i
        move = (rng.random((m, n, n)) < p_move) & (rank[:, :, None] > rank[:, None, :])  # i above j
        for r, p in enumerate(perms):
            cols = slice(r * n, (r + 1) * n)
            static[:, 0, cols] = np.arange(n, dtype=np.float32)
            for k in range(block_dim):
                static[:, 1 + k, cols] = sizes[:, :, p[k]]
            dynamic[:, 0:n, cols] = move                             # deps_move repeated per rotation
            if p[-1] != block_dim - 1:                               # vertical axis changes: needs a free side
                for sec in (1, 2):
                    side = rng.random((m, n, n)) < p_side
                    eye = np.eye(n, dtype=bool)[None] & (rng.random((m, n, 1)) < p_self)
                    side = (side & ~np.eye(n, dtype=bool)[None]) | eye
                    dynamic[:, sec * n:(sec + 1) * n, cols] = side
        return np.concatenate([static.reshape(m, -1), dynamic.reshape(m, -1)], axis=1)

    flat = _per_chunk(seed, start, batch, make)
    ns = (1 + block_dim) * n * R
    static = flat[:, :ns].reshape(batch, 1 + block_dim, n * R)
    dynamic = flat[:, ns:].reshape(batch, 3 * n, n * R)
    return torch.from_numpy(np.ascontiguousarray(static)), torch.from_numpy(np.ascontiguousarray(dynamic))


def device_rand_instances(batch, n, block_dim, seed=12345, start=0, device='cuda',
                          initial_container_width=7, initial_container_height=50):
    """RAND instances from the device-side generator (generate.generate_instances = the reference's
    generate_blocks + PACKDataset layout: random blocks packed into the initial container with
    'C+P+S-lb-hard', rejected unless all are stable, precedence extracted from the packing), drawn in
    CHUNK-sized pieces seeded by (seed, chunk index) so any CHUNK-aligned sharding sees the same data.
    -> (static, dynamic) on ``device``."""
    from . import generate
    statics, dynamics = [], []
    e = start
    while e < start + batch:
        c = e // CHUNK
        lo, hi = c * CHUNK, (c + 1) * CHUNK
        st, dy = generate.generate_instances(CHUNK, n, block_dim, initial_container_width, initial_container_height,
                                             1, (1, 5), seed=(int(seed) * 1000003 + c) % (2 ** 31), device=device)
        statics.append(st[e - lo: min(hi, start + batch) - lo])
        dynamics.append(dy[e - lo: min(hi, start + batch) - lo])
        e = hi
    return torch.cat(statics).contiguous(), torch.cat(dynamics).contiguous()


def device_ppsg_instances(batch, n, block_dim=2, seed=12345, start=0, device='cuda', initial_container_width=7,
                          initial_container_height=50, target_container_width=7, input_type='bot'):
    """Perfect-packing (PPSG) instances from the device-side generator (generate.generate_ppsg_instances_2d = the
    reference's generate_blocks_with_GT: guillotine-cut perfect packing, random take-apart order and rotations,
    hard LB_GREEDY layout in the initial container, stability and take-apart acceptance) in PACKDataset layout;
    keyed by global instance id, so any sharding sees the same data.  -> (static, dynamic) on ``device``."""
    from . import generate
    if block_dim != 2:
        raise ValueError("2D only (the 3D series is generate.generate_ppsg_instances)")
    cs = generate.initial_container(2, initial_container_width, initial_container_height)
    blocks, positions = generate.generate_ppsg_instances_2d(batch, n, initial_container_width, initial_container_height,
                                                            target_container_width, seed=seed, start=start, device=device,
                                                            input_type=input_type)
    return generate.precedence_tensors(blocks, positions, cs)


def tiled_instances(static_fix, dynamic_fix, batch, start=0):
    """``batch`` instances taken cyclically from a fixture of real instances (numpy or torch, PACKDataset
    layout): env ``start + i`` gets fixture instance ``(start + i) % len(fixture)``, so shards see what
    the single-GPU run sees."""
    st = torch.as_tensor(np.asarray(static_fix), dtype=torch.float32)
    dy = torch.as_tensor(np.asarray(dynamic_fix), dtype=torch.float32)
    idx = (torch.arange(batch) + int(start)) % st.shape[0]
    return st[idx].contiguous(), dy[idx].contiguous()


def random_feasible_tape(static, dynamic, n, seed=1, start=0):
    """A valid action tape (batch, n) int64 for the instances: at every step a uniformly random
    selectable column (numpy re-statement of the mask rule, appendix E; host only, setup time)."""
    st = static.cpu().numpy() if torch.is_tensor(static) else static
    dyn = (dynamic.cpu().numpy() if torch.is_tensor(dynamic) else dynamic).copy()
    B, rows, nR = dyn.shape
    u = _per_chunk(seed, start, B, lambda rng, m: rng.random((m, n)))
    mask = np.ones((B, nR), bool)
    tape = np.zeros((B, n), np.int64)
    ar = np.arange(B)
    for t in range(n):
        move = dyn[:, :n].sum(1); small = dyn[:, n:2 * n].sum(1); large = dyn[:, 2 * n:3 * n].sum(1)
        ok = mask & ((small * large + move) == 0)
        cnt = ok.sum(1)
        if (cnt == 0).any():
            raise ValueError("instance with no selectable block at step %d" % t)
        pick = np.minimum((u[:, t] * cnt).astype(np.int64), cnt - 1)
        csum = np.cumsum(ok, axis=1)
        ptr = (csum > pick[:, None]).argmax(1)
        tape[:, t] = ptr
        real = st[ar, 0, ptr].astype(np.int64)
        for s in range(3):
            dyn[ar, real + n * s, :] = 0
        for r in range(nR // n):
            mask[ar, (ptr % n) + n * r] = False
    return torch.from_numpy(tape)
