// tap_macs3_wave.h -- MACS / MUL 3D (tools.calc_one_position_mcs_3d, tools.py:2751-3165) for containers beyond the
// lane-per-cell kernel (more than 64 cells or a side above 8), ONE WAVEFRONT per container.  The algorithm and its state
// reduction are tap_macs3_big.h's (read that header first; it is also what the thread-per-container fallback runs): the
// control skeleton below is the same statement for statement and runs wave-uniformly, and the loops that made one
// thread's placement take milliseconds -- a single lane's ~10^4 dependent reads -- are shared by the lanes:
//   * a row of the grid (container[:, y, z] == 0, a free-list row) is ONE ballot, lane x testing its own cell;
//   * "did any cell's free-list column change at this level" is a strided OR over the cells and a wave reduction;
//   * the position table (settling level, stable) is filled one position per lane;
//   * each corner walk is a wave-wide minimum of an order rank over the positions that settle at the space's level;
//   * settled positions are scored one per lane, the first maximum by a lexicographic reduction;
//   * the tie-break (tools.py:3049-3077) tabulates, per level, the best free rectangle on either side of every row and
//     column boundary from one pass over the (first row, last row) pairs, one pair per lane (m3w_side_tables); the tied
//     candidates then sit one per lane and read the four sides of their footprint from those tables;
//   * "append if absent" compares a candidate with one list entry per lane.
// The container's working set lives in the wave's LDS tile (M3WTile).  W, L <= 64; footprints <= 8 x 8.
#pragma once

#include "tap_stable_wide.h"

#include "tap_common.h"
#include "tap_place.h"
#include "tap_macs3_big.h"

#ifdef M3W_PROF     // scratch builds only: cycles per phase of m3w_place, summed over wavefronts (scripts/m3w_phases.py)
__device__ unsigned long long m3w_prof[16];        // [0..6] cycles per phase, [8..12] counts: spaces, walks, corner hits, searches that found, placements
#define M3W_C(k, v) do { if (lane == 0) atomicAdd(&m3w_prof[k], (unsigned long long)(v)); } while (0)
#define M3W_T(k) do { const long long t_ = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(&m3w_prof[k], (unsigned long long)(t_ - t0_)); t0_ = t_; } while (0)
#else
#define M3W_T(k) do { } while (0)
#define M3W_C(k, v) do { } while (0)
#endif

struct M3WTile {                    // pointers into the wave's LDS slice
    int W, L, H, HW, flags, cap, step, n_max;
    int32_t *hm;                    // [cells]
    m3b_u64 *occ;                   // [cells][HW]
    M3BEms *ems;                    // [cap]
    int32_t *lev, *slots, *pxy;     // [cells] each; pxy[p] = px | py << 8 of position p = py * W + px (no division in the walks)
    int32_t *bs, *be;               // [64] each: m3w_side_tables' maxima by first / last row
    m3b_u64 *rows;                  // [64] scratch of the rectangle searches
    const int32_t *pos, *blk;       // history (staged in the tile by the kernel: hs = 1): entry i, coordinate k at [(i*3 + k) * hs]
    size_t hs;
};

// 8-byte units of one container's tile
__host__ __device__ inline size_t m3w_tile_u64(int cells, int HW, int n_max, int cap)
{
    return (size_t)cells * HW + 64 + (size_t)cap + ((size_t)4 * cells + 128 + 6 * (size_t)n_max + 1) / 2;
}

__device__ __forceinline__ m3b_u64 m3w_or64(m3b_u64 v)
{
    const unsigned lo = (unsigned)group_or<64>((int)(unsigned)v), hi = (unsigned)group_or<64>((int)(unsigned)(v >> 32));   // DPP + readlane (tap_place.h)
    return ((m3b_u64)hi << 32) | lo;
}
__device__ __forceinline__ int m3w_min(int v)
{
    return group_min<64>(v);
}
__device__ __forceinline__ int m3w_max(int v)
{
    return group_max<64>(v);
}

// container[:, y, z] == 0 / the free-list row (z, y) as masks over x: one ballot (every lane must call)
__device__ __forceinline__ m3b_u64 m3w_rowT(const M3WTile &s, int z, int y, int lane)
{
    return __ballot(lane < s.W && s.hm[min(lane, s.W - 1) * s.L + y] <= z);
}
__device__ __forceinline__ m3b_u64 m3w_rowF(const M3WTile &s, int z, int y, int lane)
{
    if (z < 0 || z >= s.H) return 0ull;                                          // wave-uniform
    const m3b_u64 w = ~s.occ[(size_t)(min(lane, s.W - 1) * s.L + y) * s.HW + (z >> 6)];
    return __ballot(lane < s.W && ((w >> (z & 63)) & 1ull));
}
__device__ inline int m3w_voxel(const M3WTile &s, int a, int b, int t)           // wave-uniform, as m3b_voxel
{
    if (s.hm[a * s.L + b] <= t) return 0;
    for (int q = 0; q < s.step; ++q) {
        const int bxq = s.blk[(size_t)(q * 3) * s.hs];
        if (!((bxq >> 16) & 1)) continue;
        const int z = s.pos[(size_t)(q * 3 + 2) * s.hs], zz = s.blk[(size_t)(q * 3 + 2) * s.hs];
        if (t < z || t >= z + zz) continue;
        const int x = s.pos[(size_t)(q * 3) * s.hs], y = s.pos[(size_t)(q * 3 + 1) * s.hs];
        if (a >= x && a < x + (bxq & 0xffff) && b >= y && b < y + s.blk[(size_t)(q * 3 + 1) * s.hs]) return q + 1;
    }
    return -1;
}
__device__ inline void m3w_scan(const M3WTile &s, int x, int y, int bx, int by, int &mx, m3b_u64 &eq, int &sum)
{
    mx = -1; eq = 0; sum = 0;
    for (int i = 0; i < bx; ++i)
        for (int j = 0; j < by; ++j) {
            const int h = s.hm[(x + i) * s.L + y + j];
            sum += h;
            const m3b_u64 bit = (i < 8 && j < 8) ? 1ull << (i * 8 + j) : 0ull;   // the 8 x 8 support mask; wider footprints: tap_stable_wide.h
            if (h > mx) { mx = h; eq = bit; }
            else if (h == mx) eq |= bit;
        }
}
// The tie-break's per-level tables.  A free rectangle that avoids a (filled) footprint lies wholly on one side of it, so
//   max rectangle at level h with footprint [px, px+bx) x [py, py+by) filled
//     = max( best within rows [0, px), best within rows [px+bx, n), and the same two over the other axis ),
// whether or not the footprint's cells were free.  For one axis: rows r[i] (i < n) are masks over the other axis of the
// cells with hm <= h; the (first row, last row) pairs are dealt one per lane -- (q1, q2) = (lane / (n + 1), lane % (n + 1))
// is the caller's -- and each pair's area (rows x longest common run) is maxed into bs[first] and be[last]; on return lane a
// holds PL = best rectangle within rows [0, a) and PR = best within rows [a, n) (0 from lane n on).  Every lane calls.
__device__ inline void m3w_side_tables(const M3WTile &s, int h, int n, bool over_y, int lane, int q1, int q2, int &PL, int &PR)
{
    const int L = s.L, m = over_y ? s.W : s.L;                                    // bits per row
    tap_wave_lds_sync();
    if (lane < n) {
        m3b_u64 r = 0;
        for (int b0 = 0; b0 < m; b0 += 8) {                                        // eight loads in flight (a repeated cell sets the same bit)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int b = min(b0 + j, m - 1);
                r |= (m3b_u64)(s.hm[over_y ? b * L + lane : lane * L + b] <= h) << b;
            }
        }
        s.rows[lane] = r;
    }
    s.bs[lane] = 0; s.be[lane] = 0;
    tap_wave_lds_sync();
    // the n (n + 1) / 2 pairs as ceil(n / 2) folded rows of n + 1: first rows r and n - 1 - r together have n + 1 pairs
    // (for odd n the middle row meets itself: its pairs come twice, which a maximum does not mind)
    const int fw = n + 1, d1 = 64 / fw, d2 = 64 - d1 * fw, total = ((n + 1) / 2) * fw;
    for (int idx = lane; idx < total; idx += 64) {
        const int i1 = q2 < n - q1 ? q1 : n - 1 - q1, i2 = q2 < n - q1 ? q1 + q2 : q2 - 1;   // second part: i1 + (q2 - (n - q1))
        m3b_u64 acc = ~0ull;
        for (int i0 = i1; i0 <= i2 && acc; i0 += 4) {                              // four loads in flight (a repeated row changes nothing)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc &= s.rows[min(i0 + i, i2)];
        }
        const int area = (i2 - i1 + 1) * m3b_longest_run(acc);
        if (area > 0) { atomicMax(&s.bs[i1], area); atomicMax(&s.be[i2], area); }
        q1 += d1; q2 += d2;
        if (q2 >= fw) { q2 -= fw; ++q1; }
    }
    tap_wave_lds_sync();
    int e = lane < n ? s.be[lane] : 0, b = lane < n ? s.bs[lane] : 0;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int te = __shfl_up(e, o), tb = __shfl_down(b, o);
        if (lane >= o) e = max(e, te);
        if (lane + o < 64) b = max(b, tb);
    }
    e = __shfl_up(e, 1);
    PL = lane == 0 ? 0 : e;
    PR = b;
}

// One placement; every lane of the wavefront calls it with the same arguments and gets the same result.  cnt / err as
// in m3b_place.
__device__ inline M3BResult m3w_place(const M3WTile &s, int *cnt, int &err, int bx, int by, int bz, const uint32_t *lut, int lane)
{
    const int W = s.W, L = s.L, H = s.H, HW = s.HW, cells = W * L, step = s.step;
    const bool hard = s.flags & M3B_F_HARD;
    M3BResult res = {0, 0, 0, 0, 0};
    int n_ems = 0;
    const int q1 = lane / (W + 1), q2 = lane - q1 * (W + 1), q1y = lane / (L + 1), q2y = lane - q1y * (L + 1);   // m3w_side_tables' pairs of this lane
#ifdef M3W_PROF
    long long t0_ = __builtin_readcyclecounter();
#endif
    int sx1 = 0;              // python's function-scope `x1` (tools.py:2823, 2870, 2901 assign it) ...
    bool x1def = false;       // ... which :2865 may read before any assignment (UnboundLocalError)
#define M3W_PUSH(x1_, y1_, z_, x2_, y2_)                                                                   \
    do {                                                                                                   \
        if (n_ems < s.cap) {                                                                               \
            if (lane == 0) { s.ems[n_ems].xy = (x1_) | ((y1_) << 8) | ((x2_) << 16) | ((y2_) << 24); s.ems[n_ems].z = (z_); } \
            ++n_ems;                                                                                       \
        } else err |= 16;                                                                                  \
    } while (0)
    auto absent = [&](int x1_, int y1_, int z_, int x2_, int y2_) -> bool {
        const int xy = x1_ | (y1_ << 8) | (x2_ << 16) | (y2_ << 24);
        tap_wave_lds_sync();
        bool hit = false;
        for (int i = lane; i < n_ems; i += 64) hit |= s.ems[i].xy == xy && s.ems[i].z == z_;
        return __ballot(hit) == 0ull;
    };

    // ---- (a) per-(level, row) free intervals (tools.py:2813-2841); a level whose lists all equal those of the level
    //      below is skipped (:2816), i.e. one at which no cell's column of F changes
    const int zmax = H - bz;                                                       // :2815
    for (int w = 0; w < HW && 64 * w <= zmax; ++w) {
        m3b_u64 mine = 0;
        for (int c = lane; c < cells; c += 64) {
            const m3b_u64 Fw = ~s.occ[(size_t)c * HW + w] & m3b_low(H - 64 * w);
            const m3b_u64 carry = w > 0 ? ((~s.occ[(size_t)c * HW + w - 1] & m3b_low(H - 64 * (w - 1))) >> 63) : 0ull;
            mine |= Fw ^ ((Fw << 1) | carry);
        }
        m3b_u64 chg = m3w_or64(mine) | (w == 0 ? 1ull : 0ull);
        if (zmax - 64 * w < 63) chg &= m3b_low(zmax - 64 * w + 1);
        for (; chg; chg &= chg - 1ull) {
            const int z = 64 * w + m3b_ctz(chg);
            m3b_u64 prow = 0;
            for (int y = 0; y < L; ++y) {
                if (y + by > L) break;                                             // :2818
                const m3b_u64 row = m3w_rowF(s, z, y, lane);
                const m3b_u64 prev = prow;
                prow = row;
                if (y > 0 && row == prev) continue;                                // :2819
                const m3b_u64 brow = z > 0 ? m3w_rowF(s, z - 1, y, lane) : 0ull;
                for (m3b_u64 m = row; m;) {
                    const int x1 = m3b_ctz(m), len = m3b_run_up(m, x1), x2 = x1 + len - 1;
                    const m3b_u64 run = m3b_bits(x1, x2);
                    m &= ~run;
                    sx1 = x1; x1def = true;                                        // :2823
                    if (x1 + bx > W) break;                                        // :2824
                    if (y > 0 && m3b_has_run(prev, run, x1, x2)) continue;         // :2825-2827
                    if (z > 0 && m3b_has_run(brow, run, x1, x2)) continue;         // :2828-2830
                    bool xspace = true;                                            // :2831-2840
                    int y2;
                    for (y2 = y;; ++y2) {
                        if (y2 == L - 1) break;
                        if ((m3w_rowT(s, z, y2 + 1, lane) & run) != run) break;
                        if (xspace) {
                            const m3b_u64 f = m3w_rowF(s, z, y2 + 1, lane);
                            if (!(m3b_inlist(f, x1) && m3b_inlist(f, x2))) { xspace = false; M3W_PUSH(x1, y, z, x2, y2); }
                        }
                    }
                    M3W_PUSH(x1, y, z, x2, y2);
                }
            }
        }
    }

    M3W_T(0);
    // ---- (b) spaces next to and on top of the blocks placed so far (tools.py:2843-2942); a block that could not be
    //      placed sits at (0,0,0) in `positions` and is visited all the same
    for (int bi = 0; bi < step; ++bi) {
        const int x = s.pos[(size_t)(bi * 3) * s.hs], y = s.pos[(size_t)(bi * 3 + 1) * s.hs], z = s.pos[(size_t)(bi * 3 + 2) * s.hs];
        const int xx = s.blk[(size_t)(bi * 3) * s.hs] & 0xffff, yy = s.blk[(size_t)(bi * 3 + 1) * s.hs], zz = s.blk[(size_t)(bi * 3 + 2) * s.hs];
        const int xe = x + xx - 1, t = z + zz;
        const m3b_u64 spanx = m3b_bits(x, xe);
        if (y + yy < L) {                                                          // :2847 beyond +y
            const m3b_u64 r = m3w_rowT(s, z, y + yy, lane);
            int y2;
            if ((r & spanx) == spanx) {                                            // :2849
                if (m3b_bit(r, x - 1) || (x + xx < W && m3b_bit(r, x + xx))) {
                    for (y2 = y + yy;; ++y2) { if (y2 == L - 1) break; if ((m3w_rowT(s, z, y2 + 1, lane) & spanx) != spanx) break; }
                    M3W_PUSH(x, y + yy, z, xe, y2);
                }
            } else {
                if (m3b_bit(r, x) && m3b_bit(r, x - 1)) {                          // :2858 left part
                    const int x2 = x + m3b_min(m3b_run_up(r, x), xx) - 1;          // :2860-2862
                    if (!x1def) err |= 8;                                          // :2865 UnboundLocalError
                    const m3b_u64 sp = m3b_bits(sx1, x2);                          // (sic: stale x1)
                    for (y2 = y + yy;; ++y2) { if (y2 == L - 1) break; if ((m3w_rowT(s, z, y2 + 1, lane) & sp) != sp) break; }
                    M3W_PUSH(x, y + yy, z, x2, y2);
                }
                if (m3b_bit(r, xe) && x + xx < W && m3b_bit(r, x + xx)) {          // :2868 right part
                    const int x1 = xe - m3b_min(m3b_run_down(r, xe), xx) + 1;      // :2870-2872
                    sx1 = x1; x1def = true;
                    const m3b_u64 sp = m3b_bits(x1, xe);
                    for (y2 = y + yy;; ++y2) { if (y2 == L - 1) break; if ((m3w_rowT(s, z, y2 + 1, lane) & sp) != sp) break; }
                    M3W_PUSH(x1, y + yy, z, xe, y2);
                }
            }
        }
        if (y > 0) {                                                               // :2878 beyond -y
            const m3b_u64 r = m3w_rowT(s, z, y - 1, lane);
            int y1;
            if ((r & spanx) == spanx) {
                if (m3b_bit(r, x - 1) || (x + xx < W && m3b_bit(r, x + xx))) {
                    for (y1 = y - 1;; --y1) { if (y1 == 0) break; if ((m3w_rowT(s, z, y1 - 1, lane) & spanx) != spanx) break; }
                    M3W_PUSH(x, y1, z, xe, y - 1);
                }
            } else {
                if (m3b_bit(r, x) && m3b_bit(r, x - 1)) {                          // :2889
                    const int x2 = x + m3b_min(m3b_run_up(r, x), xx) - 1;
                    const m3b_u64 sp = m3b_bits(x, x2);                            // :2896 uses x here
                    for (y1 = y - 1;; --y1) { if (y1 == 0) break; if ((m3w_rowT(s, z, y1 - 1, lane) & sp) != sp) break; }
                    M3W_PUSH(x, y1, z, x2, y - 1);
                }
                if (m3b_bit(r, xe) && x + xx < W && m3b_bit(r, x + xx)) {          // :2899
                    const int x1 = xe - m3b_min(m3b_run_down(r, xe), xx) + 1;
                    sx1 = x1; x1def = true;
                    const m3b_u64 sp = m3b_bits(x1, xe);
                    for (y1 = y - 1;; --y1) { if (y1 == 0) break; if ((m3w_rowT(s, z, y1 - 1, lane) & sp) != sp) break; }
                    M3W_PUSH(x1, y1, z, xe, y - 1);
                }
            }
        }
        if (t < H) {                                                               // :2909 on top
            m3b_u64 *tt = s.rows;                                                  // footprint rows of the top level (yy <= 16), in the wave's LDS
                                                                                   // scratch (free in this phase; every lane writes the same words and
                                                                                   // reads back what it wrote: 16 words in registers cost a wave per SIMD)
            bool full = true;
            for (int j = 0; j < yy; ++j) { tt[j] = m3w_rowT(s, t, y + j, lane); full = full && (tt[j] & spanx) == spanx; }
            if (full) {                                                            // :2911-2913
                if (absent(x, y, t, xe, y + yy - 1)) M3W_PUSH(x, y, t, xe, y + yy - 1);
            } else {                                                               // :2915-2942 partly covered
                auto hist = [&](int i, int j) -> int { return m3b_min(m3b_run_up(tt[j], x + i), xx - i) * (int)m3b_bit(tt[j], x + i); };
                auto rows_equal = [&](int i, int ja, int jb) -> bool {             // rows x+i, x+i-1 hold the same voxels over [ja, jb)
                    for (int j = ja; j < jb; ++j)
                        if (m3w_voxel(s, x + i, y + j, t) != m3w_voxel(s, x + i - 1, y + j, t)) return false;
                    return true;
                };
                for (int i = 0; i < xx; ++i)
                    for (int j = 0; j < yy; ++j) {
                        const int hv = hist(i, j);
                        if (hv == 0) continue;
                        if (j > 0 && hv == hist(i, j - 1)) continue;
                        if (i > 0 && rows_equal(i, j, yy)) continue;               // :2928
                        const int i2 = i + hv - 1;
                        int j2, j1;
                        for (j2 = j;; ++j2) { if (j2 == yy - 1) break; if (hist(i, j2 + 1) < hv) break; }
                        if (i > 0 && rows_equal(i, j, j2)) continue;               // :2934 (an empty range is "equal")
                        for (j1 = j;; --j1) { if (j1 == 0) break; if (hist(i, j1 - 1) < hv) break; }
                        if (absent(x + i, y + j1, z, x + i2, y + j2)) M3W_PUSH(x + i, y + j1, z, x + i2, y + j2);   // :2940 (sic: level z)
                    }
            }
        }
    }

    M3W_T(1);
    // ---- the four corner walks of every EMS (tools.py:3080-3115) ------------------------------------------------
    const int X = W - bx + 1, Y = L - by + 1;
    for (int p = lane; p < cells; p += 64) {                                       // position p = py * W + px, one per lane
        const int py = p / W, px = p - py * W;
        int v = -1;
        if (px < X && py < Y) {
            int mp, sum; m3b_u64 eq;
            m3w_scan(s, px, py, bx, by, mp, eq, sum);
            const int st = mp == 0 ? 1 : (bx > 8 || by > 8)
                               ? tap_stable3d_wide([&](int i, int j) { return s.hm[(px + i) * s.L + py + j]; }, bx, by, mp)
                               : tap_stable3d_any(lut, bx, by, eq);
            if (st || !hard) v = (mp << 2) | (st << 1);                            // :2963-2965
        }
        s.lev[p] = v;
        s.pxy[p] = px | (py << 8);
    }
    for (int w = 0; w < HW; ++w) {                                                 // levels at which some position settles: an EMS
        m3b_u64 mine = 0;                                                          // at any other level walks to nothing
        for (int p = lane; p < cells; p += 64) { const int v = s.lev[p]; if (v >= 0 && (v >> 8) == w) mine |= 1ull << ((v >> 2) & 63); }
        mine = m3w_or64(mine);
        if (lane == 0) s.rows[w] = mine;
    }
    tap_wave_lds_sync();
    M3W_T(2);
    int n_slots = 0;
    // one walk: the first position, in the walk's own order, of its rectangle that settles at Z and is not taken; ORDER
    // 0: x up then y up; 1: y up then x down; 2: x down then y down; 3: y down then x up
    auto walk = [&](int order, int xa, int xb, int ya, int yb, int Z) {            // rectangle [xa, xb) x [ya, yb)
        int best = INT_MAX;                                                        // rank << 12 | position (cells <= 4096)
        {   // the walk's own corner comes first in its order: when it settles there (the usual case) no search is needed
            const int pc = ((order < 2 ? ya : yb - 1) * W) + ((order == 0 || order == 3) ? xa : xb - 1), vc = s.lev[pc];
            if (vc >= 0 && !(vc & 1) && (vc >> 2) == Z) best = pc;
        }
        M3W_C(9, 1); M3W_C(10, best != INT_MAX);
        if (best == INT_MAX) {                                                     // wave-uniform
            for (int p = lane; p < cells; p += 64) {
                const int v = s.lev[p], q = s.pxy[p], px = q & 255, py = q >> 8;   // both reads go out together
                const int rank = order == 0 ? px * L + py : order == 1 ? py * W + (W - 1 - px)
                               : order == 2 ? (W - 1 - px) * L + (L - 1 - py) : (L - 1 - py) * W + px;
                const bool ok = v >= 0 && !(v & 1) && (v >> 2) == Z && px >= xa && px < xb && py >= ya && py < yb;
                best = ok ? min(best, (rank << 12) | p) : best;
            }
            best = m3w_min(best);
            M3W_C(11, best != INT_MAX);
        }
        if (best == INT_MAX) return;
        const int p = best & 4095, q = s.pxy[p];
        if (lane == 0) { s.lev[p] |= 1; s.slots[n_slots] = q; }
        ++n_slots;
        tap_wave_lds_sync();
    };
    for (int e = 0; e < n_ems; ++e) {
        const int xy = s.ems[e].xy, Z = s.ems[e].z;
        if (Z < 0 || Z >= H || !((s.rows[Z >> 6] >> (Z & 63)) & 1ull)) continue;
        const int X1 = xy & 255, Y1 = (xy >> 8) & 255, X2 = (xy >> 16) & 255, Y2 = (xy >> 24) & 255;
        const int xr = X2 - bx + 2, yr = Y2 - by + 2;                              // exclusive ends of the reversed ranges
        if (X1 < X && Y1 < Y) walk(0, X1, X, Y1, Y, Z);                            // :3085 x up, then y up
        if (xr > 0 && Y1 < Y) walk(1, 0, xr, Y1, Y, Z);                            // :3093 y up, then x down
        if (xr > 0 && yr > 0) walk(2, 0, xr, 0, yr, Z);                            // :3101 x down, then y down
        if (X1 < X && yr > 0) walk(3, X1, X, 0, yr, Z);                            // :3109 y down, then x up
    }
    // (measured and dropped: all four walks of a space from ONE pass over the positions, a walk redone on its own when an
    //  earlier walk of the space took its position -- 245 / 337 / 843 us per step at 10x10 / 12x12 / 20x20 against
    //  230 / 325 / 866 for the four passes)

    M3W_T(3);
    M3W_C(8, n_ems); M3W_C(12, 1);
    // ---- score the settled positions (tools.py:2973-2987), pick (:3118-3148) -----------------------------------
    if (n_slots == 0) return res;                                                  // :3118-3121
    int gmax = 0;
    for (int c = lane; c < cells; c += 64) gmax = max(gmax, s.hm[c]);
    gmax = m3w_max(gmax);
    const int vol = bx * by * bz, valid2 = cnt[0] + vol;
    const bool tiebreak = s.flags & M3B_F_TIE, zero = s.flags & M3B_F_ZERO;
    auto score = [&](int sl, int &px, int &py, int &mp, int &st, int &emp) -> double {
        px = s.slots[sl] & 255; py = (s.slots[sl] >> 8) & 255;
        int sum; m3b_u64 eq;
        m3w_scan(s, px, py, bx, by, mp, eq, sum);
        st = (s.lev[py * W + px] >> 1) & 1;
        emp = cnt[1] + bx * by * mp - sum;                                         // :2982-2983
        if (zero) return 0.0;                                                      // :3125
        int height = max(gmax, mp + bz);
        if (mp + bx > height) height = mp + bz;                                    // :2977 (sic block_x)
        const double C = (double)valid2 / (double)((long long)height * W * L);
        const double P = (s.flags & M3B_F_USE_P) ? (double)valid2 / (double)(emp + valid2) : 0.0;
        const double S = (s.flags & M3B_F_USE_S) ? (double)(cnt[2] + st) / (double)(cnt[3] + 1) : 0.0;
        return (C + P) + S;
    };
    // one slot per lane; first maximum in list order = lexicographic (score desc, slot asc)
    double rmax = -1.0;
    int win = INT_MAX, max_height = gmax;
    for (int sl = lane; sl < n_slots; sl += 64) {
        int px, py, mp, st, emp;
        const double r = score(sl, px, py, mp, st, emp);
        max_height = max(max_height, mp + bz);                                     // :3133
        if (r > rmax) { rmax = r; win = sl; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double r2 = __hiloint2double(__shfl_xor(__double2hiint(rmax), o), __shfl_xor(__double2loint(rmax), o));
        const int w2 = __shfl_xor(win, o);
        if (r2 > rmax || (r2 == rmax && w2 < win)) { rmax = r2; win = w2; }
    }
    max_height = m3w_max(max_height);
    int n_tied = 0;
    for (int s0 = 0; s0 < n_slots; s0 += 64) {
        const int sl = s0 + lane;
        bool tie = false;
        if (sl < n_slots) { int px, py, mp, st, emp; tie = score(sl, px, py, mp, st, emp) == rmax; }
        if (tie) s.slots[sl] |= 0x10000;                                           // marks the tie-break's candidates
        n_tied += __popcll(__ballot(tie));
    }
    M3W_T(4);
    const int nt = zero ? 4 * n_ems : n_tied;            // len(best_ems_indexes): unsettled entries score 0.0
    if (tiebreak && nt > 1) {                                                      // :3132-3144
        // calc_maximal_usable_spaces = sum over levels h < max_height of the largest free rectangle; above max(hm') a
        // level is all free, so candidates are ordered by  sum_{h < max(hm')} rect(h) - max(hm') * W * L  (tap_macs3.h).
        // Levels = 0 and the distinct heights, ascending; per level ONE set of side tables serves every candidate, the
        // candidates sit one per lane.
        if (max_height > H) err |= 1;                                              // container[:, :, h] IndexError
        long best_adj = 0;
        win = -1;
        for (int s0 = 0; s0 < n_slots; s0 += 64) {                                 // candidates in list order, 64 at a time
            const int sl = s0 + lane;
            const int q = sl < n_slots ? s.slots[sl] : 0;
            const bool mine = (q & 0x10000) != 0;
            if (__ballot(mine) == 0ull) continue;
            const int px = q & 255, py = (q >> 8) & 255;
            const int mp = mine ? s.lev[py * W + px] >> 2 : 0;
            const int Zt = mp + bz, M = max(gmax, Zt);
            const int Mtop = m3w_max(mine ? M : 0);
            long base = 0;
            for (int lo = 0; lo < Mtop;) {
                int nxt = INT_MAX;
                for (int c = lane; c < cells; c += 64) { const int hc = s.hm[c]; if (hc > lo && hc < nxt) nxt = hc; }
                nxt = m3w_min(nxt);
                int PLx, PRx, PLy, PRy;
                m3w_side_tables(s, lo, W, false, lane, q1, q2, PLx, PRx);
                m3w_side_tables(s, lo, L, true, lane, q1y, q2y, PLy, PRy);
                const int full = __shfl(PRx, 0);
                int r = max(__shfl(PLx, px), __shfl(PLy, py));
                const int rx = __shfl(PRx, min(px + bx, 63)), ry = __shfl(PRy, min(py + by, 63));
                if (px + bx < W) r = max(r, rx);
                if (py + by < L) r = max(r, ry);
                if (mine && lo < M) {
                    const int hi = min(M, nxt), a_hi = min(hi, Zt), b_lo = max(lo, Zt);
                    if (a_hi > lo) base += (long)(a_hi - lo) * r;                  // below the block's top: its footprint is filled
                    if (hi > b_lo) base += (long)(hi - b_lo) * full;
                }
                if (nxt == INT_MAX) break;
                lo = nxt;
            }
            long adj = mine ? base - (long)M * cells : LONG_MIN;
            int wsl = mine ? sl : INT_MAX;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const long a2 = ((long)__shfl_xor((int)(adj >> 32), o) << 32) | (unsigned)__shfl_xor((int)adj, o);
                const int w2 = __shfl_xor(wsl, o);
                if (a2 > adj || (a2 == adj && w2 < wsl)) { adj = a2; wsl = w2; }
            }
            if (win < 0 || adj > best_adj) { best_adj = adj; win = wsl; }
        }
    }

    M3W_T(5);
    // ---- commit (tools.py:3150-3163): every lane makes the same (idempotent) writes ----------------------------------
    {
        int px, py, Z, st, emp;
        (void)score(win, px, py, Z, st, emp);
        res.placed = 1; res.x = px; res.y = py; res.z = Z; res.stab = st;
        tap_wave_lds_sync();
        // update_level_free_space (:2989-3041) on the footprint's columns of F: levels [Z, Z+bz) are cleared; below Z a
        // row keeps its cells only when F holds the whole row AND both x-neighbours (the "strictly inside" case)
        for (int cy = py; cy < py + by; ++cy)
            for (int w = 0; w < HW; ++w) {
                m3b_u64 keep = 0;
                if (px > 0 && px + bx < W) {
                    keep = ~0ull;
                    for (int xq = px - 1; xq <= px + bx; ++xq) keep &= ~s.occ[(size_t)(xq * L + cy) * HW + w];
                }
                const m3b_u64 below = m3b_low(Z - 64 * w), upto = m3b_low(Z + bz - 64 * w);
                tap_wave_lds_sync();                                               // every lane has read the old words
                if (lane == 0)
                    for (int xq = px; xq < px + bx; ++xq)
                        s.occ[(size_t)(xq * L + cy) * HW + w] |= (upto & ~below) | (below & ~keep);
                tap_wave_lds_sync();
            }
        if (lane == 0)
            for (int i = 0; i < bx; ++i) for (int j = 0; j < by; ++j) s.hm[(px + i) * L + py + j] = Z + bz;   // :3161
        tap_wave_lds_sync();
        cnt[0] += vol;
        cnt[1] = emp;
        cnt[2] += st;
        if (Z + bz > H) err |= 1;                                                  // level_free_space[zz] IndexError
    }
    M3W_T(6);
#undef M3W_PUSH
    return res;
}
