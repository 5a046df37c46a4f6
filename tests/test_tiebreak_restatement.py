"""The two identities the wave-per-container MACS kernels' tie-break rests on (tap_macs3_wave.h: m3w_side_tables;
macs_big.hip: the tl / tr run tables), checked by brute force on small random maps.  The tie-break itself
(tools.py:3049-3077 in 3D, 2718-2736 in 2D: calc_maximal_usable_spaces of every tied candidate's map) is compared with
the oracle, end to end, by the GPU parity tests; this file pins the arithmetic the kernels replace it with.

  3D  largest all-free rectangle of a level with a candidate's footprint counted as filled
        = max over the four sides of the footprint of the largest free rectangle lying wholly on that side,
      whether or not the footprint's cells were free (a rectangle avoiding a box is separated from it by an axis).
  2D  the usable-space sum of a candidate map (one pass per column in the serial statement)
        = sum over the OLD map's levels of (longest free run left or right of the footprint) below the block's top
          and of the old map's longest run from the top up, a level counting only once some column is free."""
import numpy as np


def _maxrect(free):
    W, L = free.shape
    best = 0
    for i1 in range(W):
        acc = np.ones(L, bool)
        for i2 in range(i1, W):
            acc &= free[i2]
            run = cur = 0
            for f in acc:
                cur = cur + 1 if f else 0
                run = max(run, cur)
            best = max(best, (i2 - i1 + 1) * run)
    return best


def test_3d_rectangle_avoiding_a_footprint_lies_on_one_side():
    rs = np.random.RandomState(0)
    for _ in range(400):
        W, L = rs.randint(1, 8), rs.randint(1, 8)
        free = rs.rand(W, L) < rs.choice([0.3, 0.6, 0.9])
        bx, by = rs.randint(1, W + 1), rs.randint(1, L + 1)
        px, py = rs.randint(0, W - bx + 1), rs.randint(0, L - by + 1)
        filled = free.copy()
        filled[px:px + bx, py:py + by] = False
        sides = max(_maxrect(free[:px]) if px else 0, _maxrect(free[px + bx:]) if px + bx < W else 0,
                    _maxrect(free[:, :py]) if py else 0, _maxrect(free[:, py + by:]) if py + by < L else 0)
        assert _maxrect(filled) == sides


def _longest(mask):
    run = cur = 0
    for f in mask:
        cur = cur + 1 if f else 0
        run = max(run, cur)
    return run


def _usable_serial(hm, xs, bx, top, gmax):
    """macs_big.hip's one-thread statement (mb_adj): every distinct height v of the candidate map below m counts
    (next height - v) * (longest run of columns <= v, minus one)."""
    W, m = len(hm), max(gmax, top)
    cand = [top if xs <= k < xs + bx else hm[k] for k in range(W)]
    base = 0
    for v in sorted(set(cand)):
        if v >= m:
            continue
        nxt = min([h for h in cand if h > v] + [m])
        base += (nxt - v) * (_longest([h <= v for h in cand]) - 1)
    return base - m * (W - 1)


def _usable_tables(hm, xs, bx, top, gmax):
    W, m = len(hm), max(gmax, top)
    levels = sorted(set(hm))
    base = 0
    for k, lo in enumerate(levels):
        hi = min(levels[k + 1] if k + 1 < len(levels) else 1 << 30, m)
        free = [h <= lo for h in hm]
        side = max(_longest(free[:xs]), _longest(free[xs + bx:]))
        a_hi, b_lo = min(hi, top), max(lo, top)
        if a_hi > lo and side >= 1:
            base += (a_hi - lo) * (side - 1)
        if hi > b_lo:
            base += (hi - b_lo) * (_longest(free) - 1)
    return base - m * (W - 1)


def test_2d_usable_space_from_per_level_run_tables():
    rs = np.random.RandomState(1)
    for _ in range(3000):
        W = rs.randint(1, 14)
        hm = [int(h) for h in rs.randint(0, 7, size=W)]
        bx = rs.randint(1, W + 1)
        xs = rs.randint(0, W - bx + 1)
        top = max(hm[xs:xs + bx]) + rs.randint(1, 5)
        assert _usable_serial(hm, xs, bx, top, max(hm)) == _usable_tables(hm, xs, bx, top, max(hm))
