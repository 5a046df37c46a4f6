"""CPU checks of the synthetic generators and the sharding helpers."""
import numpy as np
import torch

import oracle_lib as O
from tap_net_amd import dist as tdist
from tap_net_amd import synth


def test_rand_blocks_marginal_and_sharding():
    b = synth.rand_blocks(4096, 10, 2, seed=7)
    assert b.shape == (4096, 10, 2) and b.min() == 1 and b.max() == 4
    freq = np.bincount(b.reshape(-1), minlength=5)[1:] / b.size
    assert np.allclose(freq, [0.15, 0.35, 0.35, 0.15], atol=0.02)
    # any contiguous shard reproduces the slice of the full batch
    lo, hi = tdist.shard_range(4096, 1, 3)
    assert np.array_equal(synth.rand_blocks(hi - lo, 10, 2, seed=7, start=lo), b[lo:hi])


def test_shard_range_covers():
    for total in (0, 1, 7, 8192, 65536):
        for world in (1, 2, 3, 8):
            spans = [tdist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_instances_are_completable_and_tape_matches_oracle_masks():
    for D in (2, 3):
        n = 6
        static, dynamic = synth.rand_instances(96, n, D, seed=3)
        R = static.shape[2] // n
        tape = synth.random_feasible_tape(static, dynamic, n, seed=5)
        st, dyn = static.numpy(), dynamic.numpy()
        cur = O.initial_mask(dyn, n)
        mask = np.ones_like(cur)
        for t in range(n):
            ptr = tape[:, t].numpy()
            assert (cur[np.arange(96), ptr] == 1).all()          # the tape only picks selectable columns
            dyn = O.update_dynamic(dyn, st, ptr, n, 3)
            cur, mask = O.update_mask(mask, dyn, ptr, n, R)
        assert not mask.any()
        # shard invariance
        s2, d2 = synth.rand_instances(40, n, D, seed=3, start=1000)
        s_full, d_full = synth.rand_instances(1100, n, D, seed=3)
        assert torch.equal(s2, s_full[1000:1040]) and torch.equal(d2, d_full[1000:1040])
