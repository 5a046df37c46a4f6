// tap_macs3_big.h -- MACS / MUL 3D (tools.calc_one_position_mcs_3d, tools.py:2751-3165) for containers the
// lane-per-cell kernel (tap_macs3.h) does not cover: more than 64 cells or a side above 8 (the reference builds
// W x W x H for any --container_width, model.py:279, and its MACS has no size limit).  ONE THREAD per container,
// walking its own state in global memory: a correctness path for unusual shapes like big.hip / macs_big.hip, not a
// fast one.
//
// Same state reduction as tap_macs3.h (read its header first): voxel (x,y,z) == 0 <=> z >= hm[x,y]; the reference's
// per-(level, row) free-interval lists are the maximal runs of a bit-grid F whose column per cell is stored
// complemented in `occ` (ceil(H/64) words per cell); voxel VALUES are only compared in the partly-covered-top case and
// are rebuilt from the placement history there; a block settles at (x, y, Z) iff Z = max of the height-map under it
// (and, for hard rewards, the position is stable), so `visited` is one flag per position.  Here rows of the grid are
// 64-bit masks over x (W, L <= 64) and every list operation is bit arithmetic on them.
// Limits: W, L <= 64; block footprints up to 16 x 16 (8 x 8: the support mask of tools.is_stable, beyond: tap_stable_wide.h); the EMS list holds
// macs3_big_cap() entries (error bit 16 beyond, like MACS3_EMS_CAP).
#pragma once

#include "tap_stable_wide.h"

#include <cstddef>
#include <cstdint>

#ifndef M3B_HD
#define M3B_HD __device__
#endif

typedef unsigned long long m3b_u64;

M3B_HD inline int m3b_ctz(m3b_u64 v) { return v ? __builtin_ctzll(v) : 64; }
M3B_HD inline int m3b_clz(m3b_u64 v) { return v ? __builtin_clzll(v) : 64; }
M3B_HD inline int m3b_min(int a, int b) { return a < b ? a : b; }
M3B_HD inline int m3b_max(int a, int b) { return a > b ? a : b; }
M3B_HD inline m3b_u64 m3b_low(int n) { return n <= 0 ? 0ull : (n >= 64 ? ~0ull : ((1ull << n) - 1ull)); }   // bits [0, n)
M3B_HD inline m3b_u64 m3b_bits(int a, int b) { return a > b ? 0ull : (m3b_low(b - a + 1) << a); }            // bits a .. b
M3B_HD inline bool m3b_bit(m3b_u64 r, int i) { return i >= 0 && i < 64 && ((r >> i) & 1ull); }
// list semantics on a row mask (tap_macs3.h: m3_has_run / m3_inlist)
M3B_HD inline bool m3b_has_run(m3b_u64 r, m3b_u64 run, int x1, int x2)
{
    return (r & run) == run && !m3b_bit(r, x1 - 1) && !m3b_bit(r, x2 + 1);
}
M3B_HD inline bool m3b_inlist(m3b_u64 r, int v) { return m3b_bit(r, v) && (v == 0 || !m3b_bit(r, v - 1) || !m3b_bit(r, v + 1)); }
M3B_HD inline int m3b_longest_run(m3b_u64 v)
{
    int r = 0;
    while (v) { v &= v << 1; ++r; }
    return r;
}
// number of consecutive set bits of r from bit x upwards / downwards (bit x included)
M3B_HD inline int m3b_run_up(m3b_u64 r, int x) { return m3b_ctz(~(r >> x)); }          // the shift brings in zeros: the run ends there
M3B_HD inline int m3b_run_down(m3b_u64 r, int x) { return m3b_clz(~(r << (63 - x))); }

constexpr int M3B_F_HARD = 1, M3B_F_USE_P = 2, M3B_F_USE_S = 4, M3B_F_ZERO = 8, M3B_F_TIE = 16;   // = TAP_F_* (tapenv.h)

struct M3BEms { int32_t xy, z; };   // xy = x1 | y1 << 8 | x2 << 16 | y2 << 24

struct M3BState {
    int W, L, H, HW, flags, cap, step;
    int32_t *hm;                    // [W*L], cell x*L + y
    m3b_u64 *occ;                   // [W*L][HW], complement of the cell's column of F
    const int32_t *pos, *blk;       // history: entry i, coordinate k at [(i*3 + k) * hs]; bit 16 of blk x = "was placed"
    size_t hs;
    M3BEms *ems;                    // [cap]
    int32_t *lev;                   // [W*L] per position y*W + x: (settling level << 2 | stable << 1 | taken), -1 = never settles
    int32_t *slots;                 // [W*L] settled positions in walk order: px | py << 8
    int32_t *lvh, *lvr;             // [n_max + 2] distinct levels of the height-map and their largest free rectangle
};

struct M3BResult { int placed, x, y, z, stab; };

M3B_HD inline m3b_u64 m3b_Fword(const M3BState &s, int cell, int w)
{
    return ~s.occ[(size_t)cell * s.HW + w] & m3b_low(s.H - 64 * w);
}
M3B_HD inline m3b_u64 m3b_rowF(const M3BState &s, int z, int y)          // free-list row (z, y) as a mask over x
{
    m3b_u64 r = 0;
    if (z < 0 || z >= s.H) return 0;
    for (int x = 0; x < s.W; ++x) r |= ((m3b_Fword(s, x * s.L + y, z >> 6) >> (z & 63)) & 1ull) << x;
    return r;
}
M3B_HD inline m3b_u64 m3b_rowT(const M3BState &s, int z, int y)          // container[:, y, z] == 0 as a mask over x
{
    m3b_u64 r = 0;
    for (int x = 0; x < s.W; ++x) r |= (m3b_u64)(s.hm[x * s.L + y] <= z) << x;
    return r;
}
// voxel value at (a, b, t): 0 free, q + 1 inside placed block q, -1 under a block (update_container, tools.py:3043-3047)
M3B_HD inline int m3b_voxel(const M3BState &s, int a, int b, int t)
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

// max height / support mask (bit i*8 + j) / sum of heights under the footprint at (x, y)
M3B_HD inline void m3b_scan(const M3BState &s, int x, int y, int bx, int by, int &mx, m3b_u64 &eq, int &sum)
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

// largest all-free rectangle at level h (cells with hm <= h), the footprint [fx, fx+fbx) x [fy, fy+fby) counted as
// filled when fbx > 0 (calc_maximal_usable_spaces' level_max, tools.py:3049-3077)
M3B_HD inline int m3b_maxrect(const M3BState &s, int h, int fx, int fy, int fbx, int fby)
{
    m3b_u64 rows[64];
    const int W = s.W, L = s.L;
    for (int x = 0; x < W; ++x) {
        m3b_u64 r = 0;
        for (int y = 0; y < L; ++y) r |= (m3b_u64)(s.hm[x * L + y] <= h) << y;
        if (fbx > 0 && x >= fx && x < fx + fbx) r &= ~m3b_bits(fy, fy + fby - 1);
        rows[x] = r;
    }
    int best = 0;
    for (int i1 = 0; i1 < W; ++i1) {
        m3b_u64 acc = ~0ull;
        for (int i2 = i1; i2 < W; ++i2) {
            acc &= rows[i2];
            if (!acc) break;
            best = m3b_max(best, (i2 - i1 + 1) * m3b_longest_run(acc));
        }
    }
    return best;
}

// The tie-break's per-level tables: PLx[a] / PRx[a] = largest all-free rectangle of level h within rows x in [0, a) /
// [a, W) (a = 0 .. W), PLy / PRy the same over y.  A free rectangle that avoids a filled footprint
// [px, px+bx) x [py, py+by) lies wholly on one side of it, so the level's largest rectangle with that footprint filled is
// max(PLx[px], PRx[px+bx], PLy[py], PRy[py+by]) whether or not the footprint's cells were free; PLx[W] is the level's own.
constexpr int M3B_TIE_CHUNK = 32;
M3B_HD inline void m3b_side_tables(const M3BState &s, int h, int *PLx, int *PRx, int *PLy, int *PRy)
{
    m3b_u64 rows[64];
    for (int axis = 0; axis < 2; ++axis) {
        const int n = axis ? s.L : s.W, m = axis ? s.W : s.L;
        int *PL = axis ? PLy : PLx, *PR = axis ? PRy : PRx;
        for (int i = 0; i < n; ++i) {
            m3b_u64 r = 0;
            for (int b = 0; b < m; ++b) r |= (m3b_u64)(s.hm[axis ? b * s.L + i : i * s.L + b] <= h) << b;
            rows[i] = r;
        }
        for (int a = 0; a <= n; ++a) { PL[a] = 0; PR[a] = 0; }
        for (int i1 = 0; i1 < n; ++i1) {
            m3b_u64 acc = ~0ull;
            for (int i2 = i1; i2 < n; ++i2) {
                acc &= rows[i2];
                if (!acc) break;
                const int area = (i2 - i1 + 1) * m3b_longest_run(acc);
                PR[i1] = m3b_max(PR[i1], area);                                    // by first row
                PL[i2 + 1] = m3b_max(PL[i2 + 1], area);                            // by last row
            }
        }
        for (int a = 1; a <= n; ++a) PL[a] = m3b_max(PL[a], PL[a - 1]);
        for (int a = n - 1; a >= 0; --a) PR[a] = m3b_max(PR[a], PR[a + 1]);
    }
}

// One placement.  STAB(bx, by, eq) = tools.is_stable on a support mask.  cnt = {valid, empty, nstable, count}; on
// return the state (hm, occ) and cnt[0..2] are updated; the caller advances cnt[3] and appends the history.
template <typename STAB>
M3B_HD inline M3BResult m3b_place(M3BState &s, int *cnt, int &err, int bx, int by, int bz, STAB stab_of)
{
    const int W = s.W, L = s.L, H = s.H, HW = s.HW, cells = W * L, step = s.step;
    const bool hard = s.flags & M3B_F_HARD;
    M3BResult res = {0, 0, 0, 0, 0};
    int n_ems = 0;
    int sx1 = 0;              // python's function-scope `x1` (tools.py:2823, 2870, 2901 assign it) ...
    bool x1def = false;       // ... which :2865 may read before any assignment (UnboundLocalError)
#define M3B_PUSH(x1_, y1_, z_, x2_, y2_)                                                                   \
    do {                                                                                                   \
        if (n_ems < s.cap) { s.ems[n_ems].xy = (x1_) | ((y1_) << 8) | ((x2_) << 16) | ((y2_) << 24); s.ems[n_ems].z = (z_); ++n_ems; } \
        else err |= 16;                                                                                    \
    } while (0)
    auto absent = [&](int x1_, int y1_, int z_, int x2_, int y2_) -> bool {
        const int xy = x1_ | (y1_ << 8) | (x2_ << 16) | (y2_ << 24);
        for (int i = 0; i < n_ems; ++i) if (s.ems[i].xy == xy && s.ems[i].z == z_) return false;
        return true;
    };

    // ---- (a) per-(level, row) free intervals (tools.py:2813-2841); a level whose lists all equal those of the level
    //      below is skipped (:2816), i.e. one at which no cell's column of F changes
    const int zmax = H - bz;                                                       // :2815
    for (int w = 0; w < HW && 64 * w <= zmax; ++w) {
        m3b_u64 chg = w == 0 ? 1ull : 0ull;
        for (int c = 0; c < cells; ++c) {
            const m3b_u64 Fw = m3b_Fword(s, c, w), carry = w > 0 ? (m3b_Fword(s, c, w - 1) >> 63) : 0ull;
            chg |= Fw ^ ((Fw << 1) | carry);
        }
        if (zmax - 64 * w < 63) chg &= m3b_low(zmax - 64 * w + 1);
        for (; chg; chg &= chg - 1ull) {
            const int z = 64 * w + m3b_ctz(chg);
            m3b_u64 prow = 0;
            for (int y = 0; y < L; ++y) {
                if (y + by > L) break;                                             // :2818
                const m3b_u64 row = m3b_rowF(s, z, y);
                const m3b_u64 prev = prow;
                prow = row;
                if (y > 0 && row == prev) continue;                                // :2819
                const m3b_u64 brow = z > 0 ? m3b_rowF(s, z - 1, y) : 0ull;
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
                        if ((m3b_rowT(s, z, y2 + 1) & run) != run) break;
                        if (xspace) {
                            const m3b_u64 f = m3b_rowF(s, z, y2 + 1);
                            if (!(m3b_inlist(f, x1) && m3b_inlist(f, x2))) { xspace = false; M3B_PUSH(x1, y, z, x2, y2); }
                        }
                    }
                    M3B_PUSH(x1, y, z, x2, y2);
                }
            }
        }
    }

    // ---- (b) spaces next to and on top of the blocks placed so far (tools.py:2843-2942); a block that could not be
    //      placed sits at (0,0,0) in `positions` and is visited all the same
    for (int bi = 0; bi < step; ++bi) {
        const int x = s.pos[(size_t)(bi * 3) * s.hs], y = s.pos[(size_t)(bi * 3 + 1) * s.hs], z = s.pos[(size_t)(bi * 3 + 2) * s.hs];
        const int xx = s.blk[(size_t)(bi * 3) * s.hs] & 0xffff, yy = s.blk[(size_t)(bi * 3 + 1) * s.hs], zz = s.blk[(size_t)(bi * 3 + 2) * s.hs];
        const int xe = x + xx - 1, t = z + zz;
        const m3b_u64 spanx = m3b_bits(x, xe);
        if (y + yy < L) {                                                          // :2847 beyond +y
            const m3b_u64 r = m3b_rowT(s, z, y + yy);
            int y2;
            if ((r & spanx) == spanx) {                                            // :2849
                if (m3b_bit(r, x - 1) || (x + xx < W && m3b_bit(r, x + xx))) {
                    for (y2 = y + yy;; ++y2) { if (y2 == L - 1) break; if ((m3b_rowT(s, z, y2 + 1) & spanx) != spanx) break; }
                    M3B_PUSH(x, y + yy, z, xe, y2);
                }
            } else {
                if (m3b_bit(r, x) && m3b_bit(r, x - 1)) {                          // :2858 left part
                    const int x2 = x + m3b_min(m3b_run_up(r, x), xx) - 1;          // :2860-2862
                    if (!x1def) err |= 8;                                          // :2865 UnboundLocalError
                    const m3b_u64 sp = m3b_bits(sx1, x2);                          // (sic: stale x1)
                    for (y2 = y + yy;; ++y2) { if (y2 == L - 1) break; if ((m3b_rowT(s, z, y2 + 1) & sp) != sp) break; }
                    M3B_PUSH(x, y + yy, z, x2, y2);
                }
                if (m3b_bit(r, xe) && x + xx < W && m3b_bit(r, x + xx)) {          // :2868 right part
                    const int x1 = xe - m3b_min(m3b_run_down(r, xe), xx) + 1;      // :2870-2872
                    sx1 = x1; x1def = true;
                    const m3b_u64 sp = m3b_bits(x1, xe);
                    for (y2 = y + yy;; ++y2) { if (y2 == L - 1) break; if ((m3b_rowT(s, z, y2 + 1) & sp) != sp) break; }
                    M3B_PUSH(x1, y + yy, z, xe, y2);
                }
            }
        }
        if (y > 0) {                                                               // :2878 beyond -y
            const m3b_u64 r = m3b_rowT(s, z, y - 1);
            int y1;
            if ((r & spanx) == spanx) {
                if (m3b_bit(r, x - 1) || (x + xx < W && m3b_bit(r, x + xx))) {
                    for (y1 = y - 1;; --y1) { if (y1 == 0) break; if ((m3b_rowT(s, z, y1 - 1) & spanx) != spanx) break; }
                    M3B_PUSH(x, y1, z, xe, y - 1);
                }
            } else {
                if (m3b_bit(r, x) && m3b_bit(r, x - 1)) {                          // :2889
                    const int x2 = x + m3b_min(m3b_run_up(r, x), xx) - 1;
                    const m3b_u64 sp = m3b_bits(x, x2);                            // :2896 uses x here
                    for (y1 = y - 1;; --y1) { if (y1 == 0) break; if ((m3b_rowT(s, z, y1 - 1) & sp) != sp) break; }
                    M3B_PUSH(x, y1, z, x2, y - 1);
                }
                if (m3b_bit(r, xe) && x + xx < W && m3b_bit(r, x + xx)) {          // :2899
                    const int x1 = xe - m3b_min(m3b_run_down(r, xe), xx) + 1;
                    sx1 = x1; x1def = true;
                    const m3b_u64 sp = m3b_bits(x1, xe);
                    for (y1 = y - 1;; --y1) { if (y1 == 0) break; if ((m3b_rowT(s, z, y1 - 1) & sp) != sp) break; }
                    M3B_PUSH(x1, y1, z, xe, y - 1);
                }
            }
        }
        if (t < H) {                                                               // :2909 on top
            m3b_u64 tt[TAP_WIDE_MAX_SIDE];                                         // footprint rows of the top level (yy <= 16)
            bool full = true;
            for (int j = 0; j < yy; ++j) { tt[j] = m3b_rowT(s, t, y + j); full = full && (tt[j] & spanx) == spanx; }
            if (full) {                                                            // :2911-2913
                if (absent(x, y, t, xe, y + yy - 1)) M3B_PUSH(x, y, t, xe, y + yy - 1);
            } else {                                                               // :2915-2942 partly covered
                auto hist = [&](int i, int j) -> int { return m3b_min(m3b_run_up(tt[j], x + i), xx - i) * (int)m3b_bit(tt[j], x + i); };
                auto rows_equal = [&](int i, int ja, int jb) -> bool {             // rows x+i, x+i-1 hold the same voxels over [ja, jb)
                    for (int j = ja; j < jb; ++j)
                        if (m3b_voxel(s, x + i, y + j, t) != m3b_voxel(s, x + i - 1, y + j, t)) return false;
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
                        if (absent(x + i, y + j1, z, x + i2, y + j2)) M3B_PUSH(x + i, y + j1, z, x + i2, y + j2);   // :2940 (sic: level z)
                    }
            }
        }
    }

    // ---- the four corner walks of every EMS (tools.py:3080-3115) ------------------------------------------------
    const int X = W - bx + 1, Y = L - by + 1;
    for (int py = 0; py < L; ++py)
        for (int px = 0; px < W; ++px) {
            int v = -1;
            if (px < X && py < Y) {
                int mp, sum; m3b_u64 eq;
                m3b_scan(s, px, py, bx, by, mp, eq, sum);
                const int st = mp == 0 ? 1 : (bx > 8 || by > 8)
                                   ? tap_stable3d_wide([&](int i, int j) { return s.hm[(px + i) * s.L + py + j]; }, bx, by, mp)
                                   : stab_of(bx, by, eq);
                if (st || !hard) v = (mp << 2) | (st << 1);                        // :2963-2965
            }
            s.lev[py * W + px] = v;
        }
    int n_slots = 0;
    auto try_pos = [&](int px, int py, int Z) -> bool {                            // check_position, :2947-2971
        const int v = s.lev[py * W + px];
        if (v < 0 || (v & 1) || (v >> 2) != Z) return false;
        s.lev[py * W + px] = v | 1;
        s.slots[n_slots++] = px | (py << 8);
        return true;
    };
    for (int e = 0; e < n_ems; ++e) {
        const int xy = s.ems[e].xy, Z = s.ems[e].z;
        const int X1 = xy & 255, Y1 = (xy >> 8) & 255, X2 = (xy >> 16) & 255, Y2 = (xy >> 24) & 255;
        const int xr = X2 - bx + 2, yr = Y2 - by + 2;                              // exclusive ends of the reversed ranges
        bool done;
        if (X1 < X && Y1 < Y) {                                                    // :3085 x up, then y up
            done = false;
            for (int _x = X1; _x < X && !done; ++_x) for (int _y = Y1; _y < Y && !done; ++_y) done = try_pos(_x, _y, Z);
        }
        if (xr > 0 && Y1 < Y) {                                                    // :3093 y up, then x down
            done = false;
            for (int _y = Y1; _y < Y && !done; ++_y) for (int _x = xr - 1; _x >= 0 && !done; --_x) done = try_pos(_x, _y, Z);
        }
        if (xr > 0 && yr > 0) {                                                    // :3101 x down, then y down
            done = false;
            for (int _x = xr - 1; _x >= 0 && !done; --_x) for (int _y = yr - 1; _y >= 0 && !done; --_y) done = try_pos(_x, _y, Z);
        }
        if (X1 < X && yr > 0) {                                                    // :3109 y down, then x up
            done = false;
            for (int _y = yr - 1; _y >= 0 && !done; --_y) for (int _x = X1; _x < X && !done; ++_x) done = try_pos(_x, _y, Z);
        }
    }

    // ---- score the settled positions (tools.py:2973-2987), pick (:3118-3148) -----------------------------------
    if (n_slots == 0) return res;                                                  // :3118-3121
    int gmax = 0;
    for (int c = 0; c < cells; ++c) gmax = m3b_max(gmax, s.hm[c]);
    const int vol = bx * by * bz, valid2 = cnt[0] + vol;
    const bool tiebreak = s.flags & M3B_F_TIE, zero = s.flags & M3B_F_ZERO;
    auto score = [&](int sl, int &px, int &py, int &mp, int &st, int &emp) -> double {
        px = s.slots[sl] & 255; py = s.slots[sl] >> 8;
        int sum; m3b_u64 eq;
        m3b_scan(s, px, py, bx, by, mp, eq, sum);
        st = (s.lev[py * W + px] >> 1) & 1;
        emp = cnt[1] + bx * by * mp - sum;                                         // :2982-2983
        if (zero) return 0.0;                                                      // :3125
        int height = m3b_max(gmax, mp + bz);
        if (mp + bx > height) height = mp + bz;                                    // :2977 (sic block_x)
        const double C = (double)valid2 / (double)((long long)height * W * L);
        const double P = (s.flags & M3B_F_USE_P) ? (double)valid2 / (double)(emp + valid2) : 0.0;
        const double S = (s.flags & M3B_F_USE_S) ? (double)(cnt[2] + st) / (double)(cnt[3] + 1) : 0.0;
        return (C + P) + S;
    };
    double rmax = -1.0;
    int win = -1, n_tied = 0, max_height = gmax;
    for (int sl = 0; sl < n_slots; ++sl) {
        int px, py, mp, st, emp;
        const double r = score(sl, px, py, mp, st, emp);
        max_height = m3b_max(max_height, mp + bz);                                 // :3133
        if (r > rmax) { rmax = r; win = sl; n_tied = 1; }                          // first maximum in list order
        else if (r == rmax) ++n_tied;
    }
    const int nt = zero ? 4 * n_ems : n_tied;            // len(best_ems_indexes): unsettled entries score 0.0
    if (tiebreak && nt > 1) {                                                      // :3132-3144
        // calc_maximal_usable_spaces = sum over levels h < max_height of the largest free rectangle; above max(hm') a
        // level is all free, so candidates are ordered by  sum_{h < max(hm')} rect(h) - max(hm') * W * L  (tap_macs3.h)
        if (max_height > H) err |= 1;                                              // container[:, :, h] IndexError
        // Per level ONE set of side tables serves every candidate (m3b_side_tables: a free rectangle that avoids the
        // candidate's filled footprint lies wholly on one side of it); the tied candidates go through the levels
        // M3B_TIE_CHUNK at a time, in list order.
        long best_adj = 0;
        win = -1;
        int PLx[65], PRx[65], PLy[65], PRy[65];
        for (int s0 = 0; s0 < n_slots;) {
            int ids[M3B_TIE_CHUNK], cpx[M3B_TIE_CHUNK], cpy[M3B_TIE_CHUNK], cZt[M3B_TIE_CHUNK], nk = 0, Mtop = 0;
            long base[M3B_TIE_CHUNK];
            for (; s0 < n_slots && nk < M3B_TIE_CHUNK; ++s0) {
                int px, py, mp, st, emp;
                if (score(s0, px, py, mp, st, emp) != rmax) continue;
                ids[nk] = s0; cpx[nk] = px; cpy[nk] = py; cZt[nk] = mp + bz; base[nk] = 0;
                Mtop = m3b_max(Mtop, m3b_max(gmax, mp + bz));
                ++nk;
            }
            if (nk == 0) break;
            for (int lo = 0; lo < Mtop;) {                                         // levels: 0 and the distinct heights, ascending
                int nxt = 0x7fffffff;
                for (int c = 0; c < cells; ++c) if (s.hm[c] > lo && s.hm[c] < nxt) nxt = s.hm[c];
                m3b_side_tables(s, lo, PLx, PRx, PLy, PRy);
                const int full = PLx[W];
                for (int k = 0; k < nk; ++k) {
                    const int Zt = cZt[k], M = m3b_max(gmax, Zt);
                    if (lo >= M) continue;
                    const int hi = m3b_min(M, nxt), a_hi = m3b_min(hi, Zt), b_lo = m3b_max(lo, Zt);
                    if (a_hi > lo)                                                 // below the block's top: its footprint is filled
                        base[k] += (long)(a_hi - lo) * m3b_max(m3b_max(PLx[cpx[k]], PRx[cpx[k] + bx]), m3b_max(PLy[cpy[k]], PRy[cpy[k] + by]));
                    if (hi > b_lo) base[k] += (long)(hi - b_lo) * full;
                }
                if (nxt == 0x7fffffff) break;
                lo = nxt;
            }
            for (int k = 0; k < nk; ++k) {
                const long adj = base[k] - (long)m3b_max(gmax, cZt[k]) * cells;
                if (win < 0 || adj > best_adj) { best_adj = adj; win = ids[k]; }
            }
        }
    }

    // ---- commit (tools.py:3150-3163) -----------------------------------------------------------------------------
    {
        int px, py, Z, st, emp;
        (void)score(win, px, py, Z, st, emp);
        res.placed = 1; res.x = px; res.y = py; res.z = Z; res.stab = st;
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
                for (int xq = px; xq < px + bx; ++xq)
                    s.occ[(size_t)(xq * L + cy) * HW + w] |= (upto & ~below) | (below & ~keep);
            }
        for (int i = 0; i < bx; ++i) for (int j = 0; j < by; ++j) s.hm[(px + i) * L + py + j] = Z + bz;   // :3161
        cnt[0] += vol;
        cnt[1] = emp;
        cnt[2] += st;
        if (Z + bz > H) err |= 1;                                                  // level_free_space[zz] IndexError
    }
#undef M3B_PUSH
    return res;
}
