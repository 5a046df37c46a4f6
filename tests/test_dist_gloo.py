"""world_size-2 gloo test of the sharding + reward all-gather path (runs on CPU)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from tap_net_amd import dist as tdist
from tap_net_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, w, _ = tdist.init_from_env(backend="gloo")
    lo, hi = tdist.shard_range(total, r, w)
    blocks = synth.rand_blocks(hi - lo, n, 2, seed=11, start=lo)
    # the env itself needs a GPU; the oracle stands in as the per-shard worker here (test only)
    desc = O.make_desc([5, 50], n, "C+P+S-lb-soft", "diff")
    local = torch.from_numpy(O.run_episodes(desc, blocks)["ratio"].astype(np.float32))
    tdist.barrier()
    full = tdist.all_gather_rewards(local, total=total)
    tmax = tdist.max_over_ranks(float(rank + 1), torch.device("cpu"))
    if r == 0:
        q.put((full.numpy(), tmax))
    dist.destroy_process_group()


def test_shard_and_all_gather_matches_single_process():
    total, n, world = 300, 6, 2          # ragged: 150/150 is even, so use 301 below too
    for total in (300, 301):
        ctx = mp.get_context("spawn")
        q = ctx.SimpleQueue()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, total, n, q)) for r in range(world)]
        [p.start() for p in procs]
        full, tmax = q.get()
        [p.join(60) for p in procs]
        assert all(p.exitcode == 0 for p in procs)
        desc = O.make_desc([5, 50], n, "C+P+S-lb-soft", "diff")
        ref = O.run_episodes(desc, synth.rand_blocks(total, n, 2, seed=11))["ratio"].astype(np.float32)
        assert np.array_equal(full, ref)
        assert tmax == 2.0
