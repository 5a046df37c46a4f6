"""Names the reference's ``tools`` module exports on the hot path (model.py:4 does ``import tools``
and builds ``tools.Container`` per env, model.py:294), backed by the HIP kernels."""
import numpy as np
import torch

from . import _lib
from .env import BatchedContainer, Container, LockstepError, lockstep_containers, lockstep_scope   # noqa: F401
from .pack import reward as _reward            # noqa: F401


def calc_positions_lb_greedy(blocks, container_size, reward_type, device='cuda'):
    """tools.calc_positions_lb_greedy (tools.py:2393-2449) for one instance:
    -> positions (n, D) int, container=None (the voxel grid is never built), stable [n] bool,
    ratio = C+P+S (un-normalised), scores = [valid, box, empty, stable_num, max_h]."""
    blocks = np.asarray(blocks).astype('int')
    n, D = blocks.shape
    env = BatchedContainer(1, container_size, n, reward_type, 'full', device=device)
    for t in range(n):
        env.add_new_blocks(torch.as_tensor(blocks[t:t + 1].astype(np.int32)), want_feature=False)
    env.check()
    cnt = env.counters[0].tolist()
    hm = env.heightmap[0].cpu().numpy()
    max_h = int(hm.max())
    box = max_h * int(np.prod(container_size[:-1]))
    valid, empty, nst = cnt[0], cnt[1], cnt[2]
    ratio = (np.float64(valid) / np.float64(box) + np.float64(valid) / np.float64(empty + valid)) \
        + np.float64(nst) / np.float64(n)
    return (env.positions[0].cpu().numpy().astype(int), None, [bool(v) for v in env.stable[0].tolist()],
            float(ratio), [valid, box, empty, nst, max_h])


def is_stable_masks(bx, by, masks, use_lut=True, device='cuda'):
    """tools.is_stable (tools.py:710-765) for a batch of support patterns of one bx x by footprint that
    is off the floor: ``masks`` (n,) integers, bit (i*by + j) = footprint cell (i, j) rests on a voxel
    (tools.py:722-728).  -> (n,) bool tensor.  ``use_lut`` picks the table the 3D kernels use for
    footprints <= 4x4, False the direct form; both are the device functions the placements call."""
    dev = _lib.resolve_device(device)
    m = torch.as_tensor(np.asarray(masks, dtype=np.uint64).view(np.int64), device=dev).contiguous()
    out = torch.empty(m.numel(), dtype=torch.uint8, device=dev)
    c = _lib.ctx(dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().tap_stable3d_eval(c, int(bx), int(by), _lib.ptr(m), m.numel(), 1 if use_lut else 0,
                                                _lib.ptr(out), _lib.stream_of(dev)), c)
    return out.bool()
