// lb.hip -- the reference's legacy packing strategy 'LB' (tools.py:3683-3686 -> calc_one_position_greedy_2d /
// _3d, tools.py:1602-1955) for B containers in lock-step.  gfx950 only.
//
// The reference itself calls this strategy abandoned (tools.py:1598-1600) but tools.Container still dispatches
// to it, so it is covered for completeness, not for speed: unlike LB_GREEDY and MACS it is NOT a function of
// the height-map -- a candidate only needs its own volume to be empty, so in soft mode blocks may float
// (tools.py:1690-1701), and two of its tests compare voxel ids of neighbouring levels (:1663, :1816-1817).
// The state is therefore the reference's own: the voxel grid (block ids, -1 under covered holes) and the
// per-level lists of free x corners with their insertion order (the walk over a list BREAKS at the first entry
// that does not fit, :1662).  One thread steps one container; everything lives in the container's slice of
// the state blob.  Quirks kept: Container never stores the returned bounding box (tools.py:3706), so
// "compactness" is valid / ((z + bz) * W * L) of the candidate alone; the spaces on top of earlier blocks are
// only tested against z + zz < H, so a block can reach above H -- the reference's list update then raises
// IndexError (:1744): error bit 1.
#include "tap_common.h"
#include "tap_place.h"

struct LbView {
    int W, L, H, D, cap;
    int16_t *vox;  // [cells][H] of this env
    uint8_t *lfs;  // [H*L][cap]
    uint8_t *lfn;  // [H*L] (list length - 1) mod 256 (so that the all-zero blob is the initial [0] everywhere; lengths 0 .. 250)
    __device__ int16_t &v(int x, int y, int z) const { return vox[(size_t)(x * L + y) * H + z]; }
    __device__ uint8_t *list(int z, int y) const { return lfs + (size_t)(z * L + y) * cap; }
    __device__ int len(int z, int y) const { return (lfn[z * L + y] + 1) & 0xff; }
    __device__ bool has(int z, int y, int x) const
    {
        const uint8_t *l = list(z, y);
        for (int i = 0, n = len(z, y); i < n; ++i) if (l[i] == x) return true;
        return false;
    }
};

struct LbBest {
    int found, x, y, z, stab, empty;
    double ratio;
};

// search one empty-maximal-space corner (tools.py:1690-1722, 1848-1879) and keep the first maximum
__device__ static void lb_try(const LbView &s, const PlaceCfg &c, const Counters &cnt, const uint32_t *lut, int ex,
                              int ey, int ez, int bx, int by, int bz, LbBest &best)
{
    const int X = s.W - bx + 1, Y = s.L - by + 1;
    const bool hard = (c.flags & TAP_F_HARD) != 0;
    for (int _x = ex; _x < X; ++_x)
        for (int _y = ey; _y < Y; ++_y) {
            bool free_ = true;                                       // numpy clips the slice at H
            for (int a = 0; a < bx && free_; ++a)
                for (int b = 0; b < by && free_; ++b)
                    for (int k = 0; k < bz && ez + k < s.H && free_; ++k) free_ = s.v(_x + a, _y + b, ez + k) == 0;
            if (!free_) continue;
            int st = 1;
            if (ez > 0) {
                u64 sup = 0;                                         // 2D: cells != 0 (:860-868); 3D: cells > 0 (:728)
                for (int a = 0; a < bx; ++a)
                    for (int b = 0; b < by; ++b) {
                        const int16_t u = s.v(_x + a, _y + b, ez - 1);
                        if (s.D == 2 ? u != 0 : u > 0) sup |= 1ull << (s.D == 2 ? a : a * 8 + b);
                    }
                if (s.D == 2) st = sup ? tap_stable2d(bx, sup) : 0;
                else st = tap_stable3d_any(lut, bx, by, sup);
            }
            if (!st && hard) continue;                               // :1698-1699, :1855-1856
            const int valid2 = cnt.valid;                            // already includes this block (:1656)
            const double C = (double)valid2 / (double)((long long)(ez + bz) * s.W * s.L);   // :1706-1711
            int emp = cnt.empty;                                     // :1713-1714
            for (int a = 0; a < bx; ++a)
                for (int b = 0; b < by; ++b)
                    for (int k = 0; k < ez; ++k) emp += s.v(_x + a, _y + b, k) == 0;
            const double P = (c.flags & TAP_F_USE_P) ? (double)valid2 / (double)(emp + valid2) : 0.0;
            const double S = (c.flags & TAP_F_USE_S) ? (double)(cnt.nstable + st) / (double)(cnt.count + 1) : 0.0;
            const double r = (C + P) + S;
            if (!best.found || r > best.ratio) { best.found = 1; best.ratio = r; best.x = _x; best.y = _y; best.z = ez; best.stab = st; best.empty = emp; }
            return;                                                  // settled: this corner is done
        }
}

__global__ void __launch_bounds__(TAP_BLOCK) k_lb_step(StepArgs a, int16_t *vox, uint8_t *lfs, uint8_t *lfn, int cap, int lpw)
{
    const int env = tap_spread_env(lpw, a.d.B);                                  // containers spread over the waves (tap_common.h)
    const int B = a.d.B;
    if (env < 0) return;
    const int D = a.d.D, W = a.d.W, L = a.d.L, H = a.d.H, cells = W * L;
    int dims[3] = {1, 1, 1};
    int err = 0;
    if (a.static_) {                                                 // gather of model.py:404-412
        bool badp;
        const long p = tap_col((long)a.ptr[env], a.nR, badp);
        for (int k = 0; k < D; ++k) dims[k] = badp ? 0 : (int)a.static_[((size_t)env * a.static_rows + 1 + k) * a.nR + p];
    } else if (a.blocks_dtype == TAP_DT_F32) {
        for (int k = 0; k < D; ++k) dims[k] = (int)((const float *)a.blocks)[(size_t)env * D + k];
    } else {
        for (int k = 0; k < D; ++k) dims[k] = ((const int32_t *)a.blocks)[(size_t)env * D + k];
    }
    const bool act = !a.active || a.active[env] != 0;
    const int4 cv = reinterpret_cast<const int4 *>(a.v.cnt)[env];
    Counters cnt = {cv.x, cv.y, cv.z, cv.w};
    const int bx = dims[0], by = D == 3 ? dims[1] : 1, bz = dims[D - 1];
    bool do_step = act;
    if (act && cnt.count >= a.d.n_max) { err |= 2; do_step = false; }
    if (act && (bx < 1 || by < 1 || bz < 1)) { err |= 4; do_step = false; }
    int32_t *hm = a.v.hm + (size_t)env * cells;
    if (do_step) {
        const LbView s = {W, L, H, D, cap, vox + (size_t)env * cells * H, lfs + (size_t)env * H * L * cap,
                          lfn + (size_t)env * H * L};
        const PlaceCfg cfg = {W, L, H, a.d.flags, a.lut};
        const int idx = cnt.count;
        cnt.valid += bx * by * bz;                                   // :1656, :1805
        LbBest best = {0, 0, 0, 0, 0, 0, 0.0};
        // corners from the level lists (:1659-1666, :1808-1820)
        for (int z = 0; z < H; ++z) {
            if (z + bz > H) break;
            if (z > 0) {
                bool any = false;
                for (int c = 0; c < cells && !any; ++c) any = s.vox[(size_t)c * H + z - 1] != 0;
                if (!any) break;
            }
            for (int y = 0; y < L; ++y) {
                if (y + by > L) break;
                if (D == 3 && y > 0 && s.len(z, y - 1) == 1 && s.list(z, y - 1)[0] == 0) continue;   // == [0]
                const uint8_t *fs = s.list(z, y);
                for (int i = 0, n = s.len(z, y); i < n; ++i) {
                    const int x = fs[i];
                    if (x + bx > W) break;
                    if (D == 3 && y > 0 && s.has(z, y - 1, x)) {
                        bool same = true;
                        for (int xx = x; xx < W && same; ++xx) same = s.v(xx, y, z) == s.v(xx, y - 1, z);
                        if (same) continue;
                    }
                    if (z > 0 && s.has(z - 1, y, x)) {
                        bool same = true;
                        for (int xx = x; xx < W && same; ++xx)
                            for (int yy = y; yy < L && same; ++yy) same = s.v(xx, yy, z) == s.v(xx, yy, z - 1);
                        if (same) continue;
                    }
                    lb_try(s, cfg, cnt, a.lut, x, y, z, bx, by, bz, best);
                }
            }
        }
        // corners on top of / behind the earlier blocks (:1668-1677, :1822-1835), unless already listed.
        // "Already listed" = produced by the loop above or by an earlier block of this loop; a corner whose
        // voxel is taken is never listed, and for equal coordinates that test gives the same answer, so it
        // is enough to look for an equal, in-range corner of an earlier block and to re-run the list test.
        auto listed = [&](int x, int y, int z) -> bool {
            if (z + bz > H || y + by > L) return false;
            for (int q = 1; q <= z; ++q) {                           // the level loop stops at the first empty level
                bool any = false;
                for (int c = 0; c < cells && !any; ++c) any = s.vox[(size_t)c * H + q - 1] != 0;
                if (!any) return false;
            }
            if (D == 3 && y > 0 && s.len(z, y - 1) == 1 && s.list(z, y - 1)[0] == 0) return false;
            const uint8_t *fs = s.list(z, y);
            bool hit = false;
            for (int i = 0, n = s.len(z, y); i < n && !hit; ++i) {
                if (fs[i] + bx > W) return false;                    // the walk breaks before reaching x
                hit = fs[i] == x;
            }
            if (!hit) return false;
            if (D == 3 && y > 0 && s.has(z, y - 1, x)) {
                bool same = true;
                for (int xx = x; xx < W && same; ++xx) same = s.v(xx, y, z) == s.v(xx, y - 1, z);
                if (same) return false;
            }
            if (z > 0 && s.has(z - 1, y, x)) {
                bool same = true;
                for (int xx = x; xx < W && same; ++xx)
                    for (int yy = y; yy < L && same; ++yy) same = s.v(xx, yy, z) == s.v(xx, yy, z - 1);
                if (same) return false;
            }
            return true;
        };
        auto corner_of = [&](int b, int which, int &x, int &y, int &z) -> bool {   // which: 0 = behind (3D), 1 = on top
            const size_t o = (size_t)b * D * B + env;
            const int px = a.v.pos[o], py = D == 3 ? a.v.pos[o + B] : 0, pz = a.v.pos[o + (size_t)(D - 1) * B];
            const int qy = D == 3 ? a.v.blk[o + B] : 1, qz = a.v.blk[o + (size_t)(D - 1) * B];
            if (which == 0) { if (D != 3 || py + qy >= L) return false; x = px; y = py + qy; z = pz; return true; }
            if (pz + qz >= H) return false;
            x = px; y = py; z = pz + qz;
            return true;
        };
        for (int b = 0; b < idx; ++b)
            for (int which = 0; which < 2; ++which) {
                int x, y, z;
                if (!corner_of(b, which, x, y, z)) continue;
                if (s.v(x, y, z) != 0) continue;
                bool dup = listed(x, y, z);
                for (int j = 0; j <= b && !dup; ++j)
                    for (int w2 = 0; w2 < 2 && !dup; ++w2) {
                        if (j == b && w2 >= which) break;
                        int x2, y2, z2;
                        dup = corner_of(j, w2, x2, y2, z2) && x2 == x && y2 == y && z2 == z;
                    }
                if (!dup) lb_try(s, cfg, cnt, a.lut, x, y, z, bx, by, bz, best);
            }
        int px = 0, py = 0, pz = 0, pst = 0;
        if (!best.found) {
            cnt.valid -= bx * by * bz;                               // :1725-1728
        } else {
            px = best.x; py = best.y; pz = best.z; pst = best.stab;
            cnt.empty = best.empty;
            cnt.nstable += pst;
            for (int i = 0; i < bx; ++i)
                for (int j = 0; j < by; ++j) {
                    for (int k = 0; k < bz && pz + k < H; ++k) s.v(px + i, py + j, pz + k) = (int16_t)(idx + 1);
                    for (int k = 0; k < pz; ++k) if (s.v(px + i, py + j, k) == 0) s.v(px + i, py + j, k) = -1;
                }
            bool over = false;
            for (int zz = 0; zz < bz && !over; ++zz)                 // :1749-1753, :1903-1907
                for (int yy = 0; yy < by; ++yy) {
                    if (pz + zz >= H) { over = true; break; }        // IndexError in the reference
                    uint8_t *l = s.list(pz + zz, py + yy);
                    int n = s.len(pz + zz, py + yy);
                    for (int i = 0; i < n; ++i)
                        if (l[i] == px) { for (int k = i; k + 1 < n; ++k) l[k] = l[k + 1]; --n; break; }
                    if (px + bx < W && s.v(px + bx, py + yy, pz + zz) == 0) {
                        if (n < cap) l[n++] = (uint8_t)(px + bx);
                        else err |= 16;          // cannot happen while the entries are distinct right edges (<= W of them)
                    }
                    s.lfn[(pz + zz) * L + py + yy] = (uint8_t)(n - 1);
                }
            if (over) err |= 1;
            else
                for (int i = 0; i < bx; ++i) for (int j = 0; j < by; ++j) hm[(px + i) * L + py + j] = pz + bz;   // :1756, :1910
        }
        const size_t o = (size_t)idx * D * B + env;
        a.v.pos[o] = px;
        if (D == 3) a.v.pos[o + B] = py;
        a.v.pos[o + (size_t)(D - 1) * B] = pz;
        a.v.blk[o] = bx;
        if (D == 3) a.v.blk[o + B] = by;
        a.v.blk[o + (size_t)(D - 1) * B] = bz;
        a.v.stable[(size_t)idx * B + env] = (uint8_t)pst;
        cnt.count += 1;                                              // tools.py:3713
        reinterpret_cast<int4 *>(a.v.cnt)[env] = make_int4(cnt.valid, cnt.empty, cnt.nstable, cnt.count);
    }
    if (err) a.v.err[env] |= err;
    if (a.feature_out) {                                             // tools.py:3716-3744
        float *out = a.feature_out + (size_t)env * a.flen;
        if (a.d.feature == TAP_FEAT_DIFF) {
            if (D == 2) { for (int c = 0; c + 1 < W; ++c) out[c] = (float)(hm[c + 1] - hm[c]); }
            else
                for (int c = 0; c < cells; ++c) {
                    const int x = c / L, y = c - x * L;
                    out[c] = (float)(x > 0 ? hm[c] - hm[c - L] : 0);
                    out[cells + c] = (float)(y > 0 ? hm[c] - hm[c - 1] : 0);
                }
        } else {
            int mn = 0;
            if (a.d.feature == TAP_FEAT_ZERO) { mn = INT_MAX; for (int c = 0; c < cells; ++c) mn = min(mn, hm[c]); }
            for (int c = 0; c < cells; ++c) out[c] = (float)(hm[c] - mn);
        }
    }
}

int tap_lb_step(tap_ctx *ctx, const StepArgs &a, void *state, hipStream_t st)
{
    if (a.d.W + 2 > 250) return tap_fail(ctx, TAP_E_UNSUPPORTED, "legacy LB: container width %d too large", a.d.W);
    if (a.d.B == 0) return TAP_OK;
    (void)state;
    const int lpw = tap_spread_lpw(a.d.B);
    hipLaunchKernelGGL(k_lb_step, dim3(tap_spread_grid(a.d.B, lpw, TAP_BLOCK)), dim3(TAP_BLOCK), 0, st, a, a.v.vox, a.v.lfs, a.v.lfn, a.d.W + 2, lpw);
    TAP_LAUNCH_CHECK(ctx, "k_lb_step");
    return TAP_OK;
}
