"""Data-parallel sharding of the env batch: one process per GPU, contiguous env ranges, no
collective on the data path; the only exchange is an all-gather of the (B/G,) fp32 reward vector
at episode end (RCCL over xGMI when the backend is "nccl"; gloo on CPU for tests)."""
import os

import torch
import torch.distributed as dist


def shard_range(total, rank, world_size):
    """Contiguous [lo, hi) of envs owned by ``rank``; sizes differ by at most one."""
    base, rem = divmod(int(total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* if launched by torchrun.
    Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    # TAP_DIST_FORCE_INIT=1: a 1-rank process group (the only way to exercise RCCL on a 1-GPU box)
    if (world > 1 or os.environ.get('TAP_DIST_FORCE_INIT') == '1') and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:   # TAP_DIST_BACKEND=gloo lets several ranks share one GPU (testing only)
            backend = os.environ.get('TAP_DIST_BACKEND') or ('nccl' if torch.cuda.is_available() else 'gloo')
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def all_gather_rewards(local_rewards, total=None):
    """All-gather per-rank reward vectors into the global (B,) vector (rank order = env order).
    Ragged shards are padded to the largest shard for the collective and trimmed afterwards."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local_rewards
    world = dist.get_world_size()
    n_local = torch.tensor([local_rewards.numel()], device=local_rewards.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    pad = torch.zeros(mx, dtype=local_rewards.dtype, device=local_rewards.device)
    pad[:local_rewards.numel()] = local_rewards
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    out = torch.cat([b[:s] for b, s in zip(bufs, sizes)])
    if total is not None and out.numel() != total:
        raise RuntimeError("gathered %d rewards, expected %d" % (out.numel(), total))
    return out


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value, device):
    """MAX-reduce a python float over ranks (timing)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
