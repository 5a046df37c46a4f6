// big.hip -- LB_GREEDY (tools.py:2027-2351) for containers the lane-per-cell kernels do not cover: more than
// 64 cells, or a 3D side above 8 (the reference builds W x W containers for any --container_width,
// model.py:279).  Same height-map formulation as tap_place.h, written for ONE THREAD per
// container walking its own cells -- a correctness path for unusual shapes, not a fast one; every BASELINE
// shape takes the lane-per-cell kernels.  Limits: W*L <= 4096 cells; 3D block footprints up to 16 x 16 (beyond the 8 x 8
// support masks: tap_stable_wide.h), larger ones raise error bit 4; 2D blocks of any width.
#include "tap_common.h"
#include "tap_place.h"
#include "tap_stable_wide.h"
#include "tap_episode.h"
#include "tap_masks.h"
#include "tap_transition.h"

// Register budget of the wave-per-container kernels.  The common placement (footprints within the support masks) needs
// 67 .. 83 VGPRs; the instantiation for wide footprints (tap_stable_wide.h: per-row extremes in 64-bit words inside the
// candidate loop) 131 .. 149 -- and a kernel is allocated the maximum over its paths, whether a wave takes them or not:
// with the wide code compiled in, the 10 x 10 step ran at 3 instead of 6 waves per SIMD, 22.9 against 19.8 us (c7,
// same-source A/B, round 5).  Holding the kernels to 5 waves per SIMD (96 VGPRs) leaves the common path unspilled and
// makes the rare wide path spill instead.
#ifndef TAP_BIG_REGS
#define TAP_BIG_REGS __attribute__((amdgpu_waves_per_eu(5, 8)))
#endif

struct BigCtx {
    int D, W, L, H, flags;
    const uint32_t *lut;
    int32_t *hm;    // this container's W*L heights (global memory)
    int32_t *keys;  // W*L ints of scratch (hard mode: candidate keys)
};

// max / support mask / sum over a footprint
// WIDE: the footprint is beyond the support mask (2D: wider than 64; 3D: a side above 8) -- maximum and sum only, the
// stability test reads the map again (big_stable).  A block's sides are the same on every lane of its container's
// wavefront, so the callers pick the instantiation once per placement: compiled into the common path (even out of line)
// the wide form cost the 10 x 10 step 15 % (c7: 23.1 against 20.1 us, round 5).
__device__ __forceinline__ bool big_is_wide(int D, int bx, int by)
{
#ifdef TAP_AB_NO_WIDE_STABLE      // A/B builds: the wide instantiation compiled out (wide blocks then get wrong stability flags)
    return false;
#else
    return D == 2 ? bx > 64 : (bx > 8 || by > 8);
#endif
}

template <bool WIDE>
__device__ static void big_scan(const BigCtx &c, int x, int y, int bx, int by, int &mx, u64 &eq, int &sum)
{
    mx = -1; eq = 0; sum = 0;
    if constexpr (WIDE) {
        for (int i = 0; i < bx; ++i)
            for (int j = 0; j < by; ++j) {
                const int h = c.hm[(x + i) * c.L + y + j];
                sum += h;
                mx = max(mx, h);
            }
        return;
    }
    for (int i = 0; i < bx; ++i)
        for (int j = 0; j < by; ++j) {
            const int h = c.hm[(x + i) * c.L + y + j];
            sum += h;
            const u64 bit = 1ull << (c.D == 2 ? i : i * 8 + j);
            if (h > mx) { mx = h; eq = bit; }
            else if (h == mx) eq |= bit;
        }
}

// the wide form out of line
__device__ __forceinline__ int big_stable_wide(const int32_t *hm, int L, int x, int y, int bx, int by, int mx)
{
    return tap_stable3d_wide([&](int i, int j) { return hm[(x + i) * L + y + j]; }, bx, by, mx);
}

// tools.is_stable_2d / is_stable of a footprint resting at level z = mx > 0 (eq: big_scan's support mask).  Beyond the
// masks' reach -- 2D blocks wider than 64, 3D sides of 9 .. 16 -- the height-map is read again (tap_stable_wide.h)
template <bool WIDE>
__device__ __forceinline__ int big_stable(const BigCtx &c, int x, int y, int bx, int by, int mx, u64 eq)
{
    if constexpr (WIDE) {
        if (c.D == 2) {
            int lead = 0, trail = 0;                                             // tools.py:839-868: leading / trailing unsupported columns
            while (lead < bx && c.hm[(x + lead) * c.L + y] != mx) ++lead;
            while (trail < bx && c.hm[(x + bx - 1 - trail) * c.L + y] != mx) ++trail;
            return (2 * lead < bx) && (2 * trail < bx);
        }
        return big_stable_wide(c.hm, c.L, x, y, bx, by, mx);
    } else {
        return c.D == 2 ? tap_stable2d(bx, eq) : tap_stable3d_any(c.lut, bx, by, eq);
    }
}

// is cell (x, y) a left-bottom corner, and of which class (tools.py:2067-2078 2D; 2219-2246 3D, appendix B)
__device__ static bool big_corner(const BigCtx &c, int x, int y, int &cls)
{
    const int L = c.L;
    cls = 0;
    if (c.D == 2) return x == 0 || c.hm[x] != c.hm[x - 1];
    if (x == 0 && y == 0) return true;
    auto HM = [&](int a, int b) { return c.hm[a * L + b]; };
    const int h = HM(x, y);
    const int hxm = x > 0 ? HM(x - 1, y) : 0, hym = y > 0 ? HM(x, y - 1) : 0;
    const int hxym = (x > 0 && y > 0) ? HM(x - 1, y - 1) : 0, hxmm = x > 1 ? HM(x - 2, y) : 0;
    const int dx = x > 0 ? h - hxm : 0, dx_ym = (x > 0 && y > 0) ? hym - hxym : 0, dx_xm = x > 1 ? hxm - hxmm : 0;
    const int dy = y > 0 ? h - hym : 0, dy_xm = (x > 0 && y > 0) ? hxm - hxym : 0;
    const bool rej1 = (y > 0) && (dx_ym != 0) && (h == hym) && (dx == dx_ym);   // :2234-2237
    const bool c1 = (dx != 0) && !rej1;
    const bool rej2 = (x > 0) && (dy_xm != 0) && (h == hxm) && (dx == dx_xm);   // :2241-2244 (sic dx)
    const bool c2 = !c1 && (dy != 0) && !rej2;
    if (c1) { cls = 1; return true; }
    if (c2) { cls = 2; return true; }
    return false;
}

__device__ static long big_key(const BigCtx &c, int x, int y, int z, int cls)
{
    return (((long)z * c.L + y) * 3 + cls) * c.W + x;                            // sort order (z, y, class, x)
}

// one placement; the caller advances the step counter.  -> placed?
template <bool WIDE>
__device__ static Placement big_place_t(const BigCtx &c, Counters &cnt, int &err, int bx, int by, int bz)
{
    const int W = c.W, L = c.L, cells = W * L;
    const bool hard = (c.flags & TAP_F_HARD) != 0;
    const PlaceCfg cfg = {W, L, c.H, c.flags, c.lut};
    const int vol = bx * by * bz;
    int gmax = 0;
    for (int i = 0; i < cells; ++i) gmax = max(gmax, c.hm[i]);
    Placement res = {0, 0, 0, 0, 0};
    int emp_w = 0;
    double best = -1.0;
    long bestkey = LONG_MAX;
    if (!hard) {
        bool stop2d = false;
        for (int x = 0; x < W && !stop2d; ++x)
            for (int y = 0; y < L; ++y) {
                int cls;
                if (!big_corner(c, x, y, cls)) continue;
                if (c.D == 2 && x + bx > W) { stop2d = true; break; }          // :2076 stops at the first overflow
                if (x + bx > W || y + by > L) continue;                          // :2255-2256
                int mx, sum; u64 eq;
                big_scan<WIDE>(c, x, y, bx, by, mx, eq, sum);
                const int z = mx;
                if (z >= c.H) err |= 1;                                          // :2109 would raise IndexError
                const int stab = z == 0 ? 1 : big_stable<WIDE>(c, x, y, bx, by, z, eq);
                const int emp = cnt.empty + bx * by * z - sum;
                const double r = tap_score(cfg, cnt, vol, gmax, z, bz, emp, stab);
                const long key = big_key(c, x, y, z, cls);
                if (r > best || (r == best && key < bestkey)) {
                    best = r; bestkey = key; res.placed = 1; res.x = x; res.y = y; res.z = z; res.stab = stab; emp_w = emp;
                }
            }
    } else {
        // keys of the in-bounds corners; then the reference's sequential walk in key order with the shared
        // visited set (tools.py:2100-2121, 2284-2297, 2320-2327) -- see tap_place for the lane-parallel form
        bool stop2d = false;
        for (int x = 0; x < W; ++x)
            for (int y = 0; y < L; ++y) {
                int cls;
                long k = LONG_MAX;
                if (!stop2d && big_corner(c, x, y, cls)) {
                    if (c.D == 2 && x + bx > W) stop2d = true;
                    else if (x + bx <= W && y + by <= L) { int mx, sum; u64 eq; big_scan<WIDE>(c, x, y, bx, by, mx, eq, sum); k = big_key(c, x, y, mx, cls); }
                }
                c.keys[x * L + y] = k > INT_MAX ? INT_MAX : (int)k;
            }
        long last = -1;
        int vis_z = -1;
        // visited spots of the current level: bit (x*L + y), cells <= 4096
        u64 visited[64];
        for (;;) {
            int kmin = INT_MAX;
            for (int i = 0; i < cells; ++i) { const int k = c.keys[i]; if ((long)k > last && k < kmin) kmin = k; }
            if (kmin == INT_MAX) break;
            last = kmin;
            const int X0 = kmin % W;
            int t = kmin / W; t /= 3;
            const int Y0 = t % L, z = t / L;
            if (z != vis_z) { vis_z = z; for (int i = 0; i < (cells + 63) / 64; ++i) visited[i] = 0; }
            bool ok = false;
            int sx = 0, sy = 0, sstab = 0, semp = 0;
            for (int _x = X0; _x + bx <= W && !ok; ++_x)
                for (int _y = Y0; _y + by <= L && !ok; ++_y) {
                    const int sp = _x * L + _y;
                    if ((visited[sp >> 6] >> (sp & 63)) & 1ull) continue;        // :2105
                    int mx, sum; u64 eq;
                    big_scan<WIDE>(c, _x, _y, bx, by, mx, eq, sum);
                    if (z > 0 && mx < z) continue;                               // :2106 nothing underneath
                    visited[sp >> 6] |= 1ull << (sp & 63);                       // :2107
                    if (z >= c.H) { err |= 1; continue; }                        // :2109 IndexError
                    if (mx > z) continue;                                        // :2109 not free
                    const int st = z == 0 ? 1 : big_stable<WIDE>(c, _x, _y, bx, by, z, eq);
                    if (!st) continue;                                           // :2112-2114
                    ok = true; sx = _x; sy = _y; sstab = st; semp = cnt.empty + bx * by * z - sum;
                }
            if (ok) {
                const double r = tap_score(cfg, cnt, vol, gmax, z, bz, semp, sstab);
                if (r > best) { best = r; res.placed = 1; res.x = sx; res.y = sy; res.z = z; res.stab = sstab; emp_w = semp; }
            }
        }
    }
    if (res.placed) {                                                            // tools.py:2167-2174
        for (int i = 0; i < bx; ++i) for (int j = 0; j < by; ++j) c.hm[(res.x + i) * L + res.y + j] = res.z + bz;
        cnt.valid += vol;
        cnt.empty = emp_w;
        cnt.nstable += res.stab;
        if (res.z + bz > c.H) err |= 1;                                          // :2169 numpy clips silently
    } else {
        res.x = res.y = res.z = res.stab = 0;
    }
    return res;
}

__device__ static Placement big_place(const BigCtx &c, Counters &cnt, int &err, int bx, int by, int bz)
{
    return big_is_wide(c.D, bx, by) ? big_place_t<true>(c, cnt, err, bx, by, bz) : big_place_t<false>(c, cnt, err, bx, by, bz);
}

__device__ static void big_feature(int feature, int D, int W, int L, const int32_t *hm, float *out)
{
    const int cells = W * L;
    if (feature == TAP_FEAT_DIFF) {                                              // tools.py:3716-3744
        if (D == 2) { for (int c = 0; c + 1 < W; ++c) out[c] = (float)(hm[c + 1] - hm[c]); }
        else
            for (int c = 0; c < cells; ++c) {
                const int x = c / L, y = c - x * L;
                out[c] = (float)(x > 0 ? hm[c] - hm[c - L] : 0);
                out[cells + c] = (float)(y > 0 ? hm[c] - hm[c - 1] : 0);
            }
        return;
    }
    int mn = 0;
    if (feature == TAP_FEAT_ZERO) { mn = INT_MAX; for (int c = 0; c < cells; ++c) mn = min(mn, hm[c]); }
    for (int c = 0; c < cells; ++c) out[c] = (float)(hm[c] - mn);
}

__global__ void __launch_bounds__(TAP_BLOCK) k_big_step(StepArgs a, int32_t *scratch, int lpw)
{
    const int env = tap_spread_env(lpw, a.d.B);                                  // containers spread over the waves (tap_common.h)
    const int B = a.d.B;
    if (env < 0) return;
    const int D = a.d.D, W = a.d.W, L = a.d.L, cells = W * L;
    int dims[3] = {1, 1, 1};
    int err = 0;
    if (a.static_) {                                                             // model.py:404-412
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
    if (act && do_step && (D == 3 && (bx > TAP_WIDE_MAX_SIDE || by > TAP_WIDE_MAX_SIDE) && bx <= W && by <= L)) {
        err |= 4; do_step = false;                                               // footprint beyond the support masks
    }
    int32_t *hm = a.v.hm + (size_t)env * cells;
    if (do_step) {
        const BigCtx c = {D, W, L, a.d.H, a.d.flags, a.lut, hm, scratch + (size_t)env * cells};
        const int step = cnt.count;
        const Placement pl = big_place(c, cnt, err, bx, by, bz);
        cnt.count += 1;                                                          // tools.py:3713
        reinterpret_cast<int4 *>(a.v.cnt)[env] = make_int4(cnt.valid, cnt.empty, cnt.nstable, cnt.count);
        int32_t *q = a.v.pos + (size_t)step * D * B + env;
        q[0] = pl.x;
        if (D == 3) { q[B] = pl.y; q[2 * (size_t)B] = pl.z; } else q[B] = pl.z;
        a.v.stable[(size_t)step * B + env] = (uint8_t)pl.stab;
    }
    if (err) a.v.err[env] |= err;
    if (a.feature_out) big_feature(a.d.feature, D, W, L, hm, a.feature_out + (size_t)env * a.flen);
}

// ---- soft rewards: one WAVEFRONT per container (round 4) ----------------------------------------------------------------
// In soft mode every corner candidate is scored independently and the winner is the first maximum in (z, y, class, x)
// order (tools.py:2161-2165, 2336-2340): lanes take the cells c = lane, lane + 64, ..., keep their best (ratio, key),
// and one wave-wide reduction on (ratio desc, key asc) picks the placement -- the container's height-map sits in the
// wave's LDS tile.  (One thread per container ran 4 096 containers as 64 wavefronts, each the union of 64 control
// flows: 265 us per step at 10 x 10 x 50, 1.9 ms with hard rewards.)  The thread-per-container kernel above remains for
// containers whose tiles do not fit a workgroup's LDS (hard rewards above 2 560 cells).
// HARD: the tile also holds, per cell, the corner key and the position's (max height under the footprint, stable) pair
// and height sum: 4 ints per cell instead of 1.
// One LB_GREEDY placement of a container whose height-map sits in the wave's LDS tile `hm` (HARD: 4 ints per cell, see
// k_big_wave_step): every lane of the wavefront calls this.  Updates the tile (and the global copy `ghm` when given),
// valid / empty / stable in `cnt` and the error bits; the caller advances cnt.count and files the result.  The returned
// placement is the same on every lane.
template <bool HARD, bool WIDE>
__device__ __forceinline__ Placement big_wave_place_t(const tap_env_desc &d, const uint32_t *lut, int32_t *hm, int32_t *ghm, int lane,
                                                      int gmax, Counters &cnt, int &err, int bx, int by, int bz)
{
    const int D = d.D, W = d.W, L = d.L, cells = W * L;
    const BigCtx c = {D, W, L, d.H, d.flags, lut, hm, nullptr};
    const PlaceCfg cfg = {W, L, d.H, d.flags, lut};
    const int vol = bx * by * bz;
    double best = -1.0;
    long bestkey = LONG_MAX;
    int bxy = 0, bzv = 0, bstab = 0, bemp = 0;
    double wr;
    long wk;
    if (!HARD) {
    for (int cell = lane; cell < cells; cell += 64) {
        const int x = cell / L, y = cell - x * L;
        int cls;
        if (!big_corner(c, x, y, cls)) continue;
        if (x + bx > W || y + by > L) continue;                              // :2076 (2D: every later corner overflows too), :2255-2256
        int mx, sum; u64 eq;
        big_scan<WIDE>(c, x, y, bx, by, mx, eq, sum);
        const int z = mx;
        if (z >= d.H) err |= 1;                                            // :2109 would raise IndexError
        const int stab = z == 0 ? 1 : big_stable<WIDE>(c, x, y, bx, by, z, eq);
        const int emp = cnt.empty + bx * by * z - sum;
        const double r = tap_score(cfg, cnt, vol, gmax, z, bz, emp, stab);
        const long key = big_key(c, x, y, z, cls);
        if (r > best || (r == best && key < bestkey)) { best = r; bestkey = key; bxy = x | (y << 12); bzv = z; bstab = stab; bemp = emp; }
    }
    // first maximum in key order over the wave
    wr = best;
    wk = bestkey;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double r2 = __hiloint2double(__shfl_xor(__double2hiint(wr), o), __shfl_xor(__double2loint(wr), o));
        const long k2 = ((long)__shfl_xor((int)(wk >> 32), o) << 32) | (unsigned)__shfl_xor((int)wk, o);
        if (r2 > wr || (r2 == wr && k2 < wk)) { wr = r2; wk = k2; }
    }
    } else {
        // The reference walks the corners in key order and slides each one (x up, then y up from the corner) to the
        // first position that is supported, free and stable, skipping positions an earlier walk of the same level
        // visited (tools.py:2100-2121, 2284-2297, 2320-2327).  A position settles at level z iff the maximum under its
        // footprint IS z and it is stable, so it can settle at one level only and "visited" only matters for
        // positions that settled: per position (max, stable, sum) once, then per corner one wave-wide minimum over
        // the order index x * L + y of the settling, untaken positions of its rectangle.
        int32_t *keys = hm + cells, *pms = keys + cells, *psum = pms + cells;
        for (int cell = lane; cell < cells; cell += 64) {
            const int x = cell / L, y = cell - x * L;
            int cls, k = INT_MAX, ms = -1, sm = 0;
            if (x + bx <= W && y + by <= L) {
                int mx; u64 eq;
                big_scan<WIDE>(c, x, y, bx, by, mx, eq, sm);
                const int st = mx == 0 ? 1 : big_stable<WIDE>(c, x, y, bx, by, mx, eq);
                ms = (mx << 1) | st;
                if (big_corner(c, x, y, cls)) { const long kk = big_key(c, x, y, mx, cls); k = kk > INT_MAX ? INT_MAX : (int)kk; }
            }
            keys[cell] = k; pms[cell] = ms; psum[cell] = sm;
        }
        tap_wave_lds_sync();
        wr = -1.0; wk = 0;
        int last = -1;
        for (;;) {
            int kmin = INT_MAX;
            for (int cell = lane; cell < cells; cell += 64) { const int k = keys[cell]; if (k > last && k < kmin) kmin = k; }
            kmin = group_min<64>(kmin);
            if (kmin == INT_MAX) break;
            last = kmin;
            const int X0 = kmin % W;
            int t = kmin / W; t /= 3;
            const int Y0 = t % L, z = t / L;
            if (z >= d.H) { err |= 1; continue; }                          // :2109 IndexError at the corner's own position
            int first = INT_MAX;
            for (int cell = lane; cell < cells; cell += 64) {
                const int x = cell / L, y = cell - x * L;
                const int ms = pms[cell];
                if (x >= X0 && y >= Y0 && ms == ((z << 1) | 1) && cell < first) first = cell;   // ms < 0: out of bounds or taken
            }
            first = group_min<64>(first);
            if (first == INT_MAX) continue;
            if (lane == 0) pms[first] = -2;                                  // settled: no later walk stops here
            const int sx = first / L, sy = first - sx * L, semp = cnt.empty + bx * by * z - psum[first];
            const double r = tap_score(cfg, cnt, vol, gmax, z, bz, semp, 1);
            if (r > wr) { wr = r; bxy = sx | (sy << 12); bzv = z; bstab = 1; bemp = semp; }    // first maximum in walk order
            tap_wave_lds_sync();
        }
        best = wr; bestkey = wk;                                             // every lane holds the winner
    }
    const bool placed = wr > 0.0;
    const u64 wm = __ballot(placed && best == wr && bestkey == wk);          // keys are unique: one lane
    const int src = wm ? __ffsll((long long)wm) - 1 : 0;
    const int pxy = __shfl(bxy, src), pz = __shfl(bzv, src), pstab = __shfl(bstab, src), pemp = __shfl(bemp, src);
    const int px = pxy & 4095, py = pxy >> 12;
    unsigned eall = (unsigned)err;
    eall = (unsigned)group_or<64>((int)eall);
    err = (int)eall;
    tap_wave_lds_sync();
    if (placed) {                                                            // tools.py:2167-2174
        for (int f = lane; f < bx * by; f += 64) {
            const int i = f / by, j = f - i * by;
            const int cidx = (px + i) * L + py + j;
            hm[cidx] = pz + bz;
            if (ghm) ghm[cidx] = pz + bz;
        }
        cnt.valid += vol;
        cnt.empty = pemp;
        cnt.nstable += pstab;
        if (pz + bz > d.H) err |= 1;                                       // :2169 numpy clips silently
    }
    Placement res = {0, 0, 0, 0, 0};
    if (placed) { res.placed = 1; res.x = px; res.y = py; res.z = pz; res.stab = pstab; }
    return res;
}

template <bool HARD>
__device__ __forceinline__ Placement big_wave_place(const tap_env_desc &d, const uint32_t *lut, int32_t *hm, int32_t *ghm, int lane,
                                                    int gmax, Counters &cnt, int &err, int bx, int by, int bz)
{
    if (__builtin_expect(big_is_wide(d.D, bx, by), 0))                            // wave-uniform, rare
        return big_wave_place_t<HARD, true>(d, lut, hm, ghm, lane, gmax, cnt, err, bx, by, bz);
    return big_wave_place_t<HARD, false>(d, lut, hm, ghm, lane, gmax, cnt, err, bx, by, bz);
}

// One lock-step of one container by one wavefront (every lane calls): gather / block, placement, state and results out,
// feature.  `hm` = the wave's LDS tile.  flags (TAP_T_FRESH: the step starts from an empty container; TAP_T_RATIO: emit
// calc_ratio, tools.py:3887-3966) as in the fused lane-per-cell step; the gather's by-products (tap_step_aux) are written
// when the step gathers.
template <bool HARD>
__device__ __forceinline__ void big_wave_step_body(const StepArgs &a, int env, int lane, int32_t *hm, int flags, float *ratio_out)
{
    const int B = a.d.B, D = a.d.D, W = a.d.W, L = a.d.L, cells = W * L;
    const bool fresh = (flags & TAP_T_FRESH) != 0;
    int32_t *ghm = a.v.hm + (size_t)env * cells;
    int gmax = 0;
    if (fresh) { for (int c = lane; c < cells; c += 64) { hm[c] = 0; ghm[c] = 0; } }
    else for (int c = lane; c < cells; c += 64) { const int h = ghm[c]; hm[c] = h; gmax = max(gmax, h); }
    int dims[3] = {1, 1, 1};
    float fv[3] = {0.f, 0.f, 0.f};
    long praw = 0;
    if (a.static_) {                                                             // model.py:404-412
        bool badp;
        praw = (long)a.ptr[env];
        const long p = tap_col(praw, a.nR, badp);
        for (int k = 0; k < D; ++k) {
            const float v = a.static_[((size_t)env * a.static_rows + 1 + k) * a.nR + p];
            dims[k] = badp ? 0 : (int)v;
            fv[k] = badp ? 0.f : v;
        }
    } else if (a.blocks_dtype == TAP_DT_F32) {
        for (int k = 0; k < D; ++k) dims[k] = (int)((const float *)a.blocks)[(size_t)env * D + k];
    } else {
        for (int k = 0; k < D; ++k) dims[k] = ((const int32_t *)a.blocks)[(size_t)env * D + k];
    }
    const bool act = !a.active || a.active[env] != 0;
    const int4 cv = reinterpret_cast<const int4 *>(a.v.cnt)[env];
    Counters cnt = {cv.x, cv.y, cv.z, cv.w};
    if (fresh) cnt = Counters{0, 0, 0, 0};
    const int bx = dims[0], by = D == 3 ? dims[1] : 1, bz = dims[D - 1];
    int err = 0;
    bool do_step = act;
    if (act && cnt.count >= a.d.n_max) { err |= 2; do_step = false; }
    if (act && (bx < 1 || by < 1 || bz < 1)) { err |= 4; do_step = false; }
    if (act && do_step && (D == 3 && (bx > TAP_WIDE_MAX_SIDE || by > TAP_WIDE_MAX_SIDE) && bx <= W && by <= L)) {
        err |= 4; do_step = false;                                               // footprint beyond the support masks
    }
    gmax = group_max<64>(gmax);
    tap_wave_lds_sync();
    if (do_step) {                                                               // wave-uniform
        const int step = cnt.count;
        const Placement pl = big_wave_place<HARD>(a.d, a.lut, hm, ghm, lane, gmax, cnt, err, bx, by, bz);
        if (pl.placed) gmax = max(gmax, pl.z + bz);
        cnt.count += 1;                                                          // tools.py:3713
        if (lane == 0) {
            int32_t *q = a.v.pos + (size_t)step * D * B + env;
            q[0] = pl.x;
            if (D == 3) { q[B] = pl.y; q[2 * (size_t)B] = pl.z; } else q[B] = pl.z;
            a.v.stable[(size_t)step * B + env] = (uint8_t)pl.stab;
        }
        tap_wave_lds_sync();
    }
    if (lane == 0) {
        if (do_step || fresh) reinterpret_cast<int4 *>(a.v.cnt)[env] = make_int4(cnt.valid, cnt.empty, cnt.nstable, cnt.count);
        if (fresh) a.v.err[env] = err;
        else if (err) a.v.err[env] |= err;
        if (a.static_) tap_step_aux(a, env, D, fv, praw);
        if ((flags & TAP_T_RATIO) && ratio_out) {                                // tools.py:3887-3966 on the state just written
            double Cc = 0.0, P = 0.0, S = 0.0;
            if (cnt.count != 0) {
                Cc = (double)cnt.valid / (double)((long long)W * L * gmax);
                P = (double)cnt.valid / (double)(cnt.empty + cnt.valid);
                S = (double)cnt.nstable / (double)cnt.count;
            }
            ratio_out[env] = (float)tap_ratio_formula(a.d.ratio_mode, Cc, P, S);
        }
    }
    if (a.feature_out) {                                                         // tools.py:3716-3744, lanes over the cells
        float *out = a.feature_out + (size_t)env * a.flen;
        if (a.d.feature == TAP_FEAT_DIFF) {
            if (D == 2) { for (int c = lane; c + 1 < W; c += 64) out[c] = (float)(hm[c + 1] - hm[c]); }
            else
                for (int c = lane; c < cells; c += 64) {
                    const int x = c / L, y = c - x * L;
                    out[c] = (float)(x > 0 ? hm[c] - hm[c - L] : 0);
                    out[cells + c] = (float)(y > 0 ? hm[c] - hm[c - 1] : 0);
                }
        } else {
            int mn = 0;
            if (a.d.feature == TAP_FEAT_ZERO) {
                mn = INT_MAX;
                for (int c = lane; c < cells; c += 64) mn = min(mn, hm[c]);
                mn = group_min<64>(mn);
            }
            for (int c = lane; c < cells; c += 64) out[c] = (float)(hm[c] - mn);
        }
    }
}

template <bool HARD>
__global__ void __launch_bounds__(TAP_BLOCK) TAP_BIG_REGS k_big_wave_step(StepArgs a)
{
    extern __shared__ int32_t big_lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;   // (a vector value: TAP_WAVE_INDEX() measured slower / flat here, tap_common.h)
    const int env = (int)(blockIdx.x * (blockDim.x >> 6)) + wave;               // 1 .. 4 wavefronts per workgroup (tap_big_step)
    if (env >= a.d.B) return;                                                     // wave-uniform
    big_wave_step_body<HARD>(a, env, lane, big_lds + (size_t)wave * a.d.W * a.d.L * (HARD ? 4 : 1), 0, nullptr);
}

// ---- the whole decoding step in ONE launch (round 5): a workgroup = PW placement wavefronts (one container each, as
// above) + ceil(PW / 2) stream waves running update_dynamic + update_mask of the same containers on the bit shadow
// (tap_transition.h: trans_stream_wave), side by side as in k_transition -- the two kinds exchange nothing, so there is
// no barrier.  MODE 1: on the bit shadow; 2: the episode's first step (shadow built in the launch).  Round 4 ran these
// shapes as a mask launch + a placement launch (+ reset / calc_ratio launches at the ends of an episode): 10 x 10 x 50,
// B = 4 096, graph-replayed step 19.5 us (c7).  Measured against it: the container's own wavefront running its slab
// first and then the placement (no stream waves: 12 instead of 8 placement waves per CU at this kernel's 141 VGPRs)
// -- 25.0 us: the placement is short enough here that the serial slab shows; the MACS forms (macs_big.hip,
// macs3_big.hip), whose placements last 40 .. 150 us, are built that way.
template <bool HARD, int NC, int MODE>
__global__ void __launch_bounds__(384) TAP_BIG_REGS k_big_transition(TransArgs a, int PW)
{
    extern __shared__ int32_t big_lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;   // (a vector value: TAP_WAVE_INDEX() measured slower / flat here, tap_common.h)
    const int cells = a.s.d.W * a.s.d.L, tile = cells * (HARD ? 4 : 1);
    const int base = blockIdx.x * PW;
    if (wave < PW) {
        const int env = base + wave;
        if (env >= a.s.d.B) return;                                               // wave-uniform
        __builtin_amdgcn_s_setprio(2);
        big_wave_step_body<HARD>(a.s, env, lane, big_lds + (size_t)wave * tile, a.flags, a.ratio_out);
        return;
    }
    const int sw = wave - PW;
    MaskArgs m = a.m;
    m.B = min(m.B, base + PW);                                                    // this workgroup's containers only
    float *slds = reinterpret_cast<float *>(big_lds + (size_t)PW * tile) + (size_t)sw * 2 * 3 * m.nR;
    trans_stream_wave<2, NC, MODE>(m, base + 2 * sw, lane, slds);
}

// ---- whole episodes (round 5): tools.calc_positions_lb_greedy (tools.py:2393-2449) for containers above 64 cells ----
// pack.reward, pack.render and -- in hard mode, inside their acceptance loops -- both instance generators
// (generate.py:908, 112) pack a whole block list from an empty container.  One wavefront per container as in the step
// kernel, the height-map tile living in LDS across the n placements: nothing but the block list and the per-episode
// results touches memory (round 4 stepped these shapes with n launches from the host, and generate.pack_blocks
// refused them).  A tour entry's block is fetched by lane t % 64 for 64 steps at a time (two dependent loads once per
// 64 placements instead of once per placement).
template <bool HARD>
__global__ void __launch_bounds__(TAP_BLOCK) TAP_BIG_REGS k_big_wave_episode(EpisodeArgs a)
{
    extern __shared__ int32_t big_lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;   // (a vector value: TAP_WAVE_INDEX() measured slower / flat here, tap_common.h)
    const int env = (int)(blockIdx.x * (blockDim.x >> 6)) + wave;
    const int D = a.d.D, W = a.d.W, L = a.d.L, cells = W * L, n = a.n;
    if (env >= a.B) return;                                                       // wave-uniform
    int32_t *hm = big_lds + (size_t)wave * cells * (HARD ? 4 : 1);
    for (int c = lane; c < cells; c += 64) hm[c] = 0;
    tap_wave_lds_sync();
    Counters cnt = {0, 0, 0, 0};
    int err = 0, gmax = 0;
    for (int t0 = 0; t0 < n; t0 += 64) {
        int mine[3] = {1, 1, 1}, merr = 0;
        bool min_ = false;
        if (t0 + lane < n) {
            if (D == 2) min_ = episode_block<2>(a, env, t0 + lane, true, mine, merr);
            else min_ = episode_block<3>(a, env, t0 + lane, true, mine, merr);
        }
        err |= merr;                                                              // OR-ed over the wave in big_wave_place / below
        for (int j = 0; j < 64 && t0 + j < n; ++j) {
            const int t = t0 + j;
            const int b0 = __shfl(mine[0], j), b1 = __shfl(mine[1], j), b2 = __shfl(mine[2], j);
            const bool in = __shfl((int)min_, j) != 0;
            const int bx = b0, by = D == 3 ? b1 : 1, bz = D == 3 ? b2 : b1;
            bool do_step = in;
            if (in && cnt.count >= a.d.n_max) { err |= 2; do_step = false; }
            if (in && (bx < 1 || by < 1 || bz < 1)) { err |= 4; do_step = false; }
            if (do_step && (D == 3 && (bx > TAP_WIDE_MAX_SIDE || by > TAP_WIDE_MAX_SIDE) && bx <= W && by <= L)) {
                err |= 4; do_step = false;                                        // footprint beyond the support masks
            }
            Placement pl = {0, 0, 0, 0, 0};
            if (do_step) {                                                        // wave-uniform
                pl = big_wave_place<HARD>(a.d, a.lut, hm, nullptr, lane, gmax, cnt, err, bx, by, bz);
                if (pl.placed) gmax = max(gmax, pl.z + bz);
                cnt.count += 1;                                                   // tools.py:3713
                tap_wave_lds_sync();
            }
            if (lane == 0) {
                if (a.pos_out) {
                    int32_t *pp = a.pos_out + ((size_t)env * n + t) * D;
                    pp[0] = pl.x;
                    if (D == 3) { pp[1] = pl.y; pp[2] = pl.z; } else pp[1] = pl.z;
                }
                if (a.stable_out) a.stable_out[(size_t)env * n + t] = (uint8_t)pl.stab;
            }
        }
    }
    err = group_or<64>(err);
    if (lane == 0) episode_finish(a, env, cnt, gmax, err);
}

int tap_big_episode(tap_ctx *ctx, const EpisodeArgs &a, hipStream_t st)
{
    if (a.B == 0) return TAP_OK;
    const bool hard = (a.d.flags & TAP_F_HARD) != 0;
    const size_t tile = (size_t)a.d.W * a.d.L * sizeof(int32_t) * (hard ? 4 : 1);
    int waves = TAP_BLOCK / 64;
    while (waves > 1 && (size_t)waves * tile > tap_lds_limit(ctx)) waves >>= 1;
    const size_t lds = (size_t)waves * tile;
    if (lds > tap_lds_limit(ctx) || tap_wave_kernels_off())
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "whole episodes of %d x %d containers%s: the height-map tile does not fit a workgroup's "
                                                "LDS, step them with tap_env_step_gather", a.d.W, a.d.L, hard ? " with hard rewards" : "");
    const dim3 g((a.B + waves - 1) / waves);
    if (hard) {
        TAP_HIP_CHECK(ctx, tap_allow_lds(k_big_wave_episode<true>, lds));
        hipLaunchKernelGGL(k_big_wave_episode<true>, g, dim3(waves * 64), lds, st, a);
    } else {
        TAP_HIP_CHECK(ctx, tap_allow_lds(k_big_wave_episode<false>, lds));
        hipLaunchKernelGGL(k_big_wave_episode<false>, g, dim3(waves * 64), lds, st, a);
    }
    TAP_LAUNCH_CHECK(ctx, "k_big_wave_episode");
    return TAP_OK;
}

// get_heightmap's feature of the current maps (tools.py:3716-3744), one wavefront per container, lanes over the cells
// (it was one thread per container until the end of round 4: 74 .. 340 us per call at 10x10 .. 20x20, more than the
// MACS 3D placement it follows)
__global__ void __launch_bounds__(TAP_BLOCK) k_big_feature(tap_env_desc d, EnvView v, float *out, int flen)
{
    const int lane = threadIdx.x & 63;
    const int env = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (env >= d.B) return;                                                       // wave-uniform
    const int W = d.W, L = d.L, cells = W * L;
    const int32_t *hm = v.hm + (size_t)env * cells;
    float *o = out + (size_t)env * flen;
    if (d.feature == TAP_FEAT_DIFF) {
        if (d.D == 2) { for (int c = lane; c + 1 < W; c += 64) o[c] = (float)(hm[c + 1] - hm[c]); }
        else {
            int x = lane / L, y = lane - x * L;
            const int dx = 64 / L, dy = 64 - dx * L;
            for (int c = lane; c < cells; c += 64) {
                o[c] = (float)(x > 0 ? hm[c] - hm[c - L] : 0);
                o[cells + c] = (float)(y > 0 ? hm[c] - hm[c - 1] : 0);
                x += dx; y += dy;
                if (y >= L) { y -= L; ++x; }
            }
        }
        return;
    }
    int mn = 0;
    if (d.feature == TAP_FEAT_ZERO) {
        mn = INT_MAX;
        for (int c = lane; c < cells; c += 64) mn = min(mn, hm[c]);
        mn = group_min<64>(mn);
    }
    for (int c = lane; c < cells; c += 64) o[c] = (float)(hm[c] - mn);
}

int tap_big_step(tap_ctx *ctx, const StepArgs &a, void *state, hipStream_t st)
{
    const int grid = (a.d.B + TAP_BLOCK - 1) / TAP_BLOCK;
    if (grid == 0) return TAP_OK;
    (void)state;
    const bool hard = (a.d.flags & TAP_F_HARD) != 0;
    const size_t tile = (size_t)a.d.W * a.d.L * sizeof(int32_t) * (hard ? 4 : 1);
    int waves = TAP_BLOCK / 64;                                                  // per workgroup: as many as the LDS holds tiles for
    while (waves > 1 && (size_t)waves * tile > tap_lds_limit(ctx)) waves >>= 1;
    const size_t lds = (size_t)waves * tile;
    if (lds <= tap_lds_limit(ctx) && !tap_wave_kernels_off()) {                  // one wavefront per container
        const dim3 g((a.d.B + waves - 1) / waves);
        if (hard) {
            TAP_HIP_CHECK(ctx, tap_allow_lds(k_big_wave_step<true>, lds));
            hipLaunchKernelGGL(k_big_wave_step<true>, g, dim3(waves * 64), lds, st, a);
        } else {
            TAP_HIP_CHECK(ctx, tap_allow_lds(k_big_wave_step<false>, lds));
            hipLaunchKernelGGL(k_big_wave_step<false>, g, dim3(waves * 64), lds, st, a);
        }
        TAP_LAUNCH_CHECK(ctx, "k_big_wave_step");
        return TAP_OK;
    }
    const int lpw = tap_spread_lpw(a.d.B);
    hipLaunchKernelGGL(k_big_step, dim3(tap_spread_grid(a.d.B, lpw, TAP_BLOCK)), dim3(TAP_BLOCK), 0, st, a, a.v.scratch, lpw);
    TAP_LAUNCH_CHECK(ctx, "k_big_step");
    return TAP_OK;
}

// placement wavefronts per workgroup of the fused step for this shape, 0 = no fused kernel (the tile does not fit)
static int big_transition_pw(const tap_ctx *ctx, const tap_env_desc *d, int nR)
{
    if (tap_wave_kernels_off() || d->strategy != TAP_LB_GREEDY) return 0;
    const bool hard = (d->flags & TAP_F_HARD) != 0;
    const size_t tile = (size_t)d->W * d->L * sizeof(int32_t) * (hard ? 4 : 1);
    for (int pw = 4; pw >= 1; pw >>= 1)
        if ((size_t)pw * tile + (size_t)((pw + 1) / 2) * 2 * 3 * nR * sizeof(float) <= tap_lds_limit(ctx)) return pw;
    return 0;
}

bool tap_big_transition_ok(const tap_ctx *ctx, const tap_env_desc *d, int nR) { return big_transition_pw(ctx, d, nR) > 0; }

int tap_big_transition(tap_ctx *ctx, const tap_env_desc *d, const TransArgs &a, hipStream_t st)
{
    const int pw = big_transition_pw(ctx, d, a.m.nR);
    if (pw == 0) return tap_fail(ctx, TAP_E_UNSUPPORTED, "no fused step for this container");
    const int mode = a.m.bits_in ? 1 : 2;
    const bool hard = (d->flags & TAP_F_HARD) != 0;
    const size_t tile = (size_t)d->W * d->L * sizeof(int32_t) * (hard ? 4 : 1);
    const int sw = (pw + 1) / 2;
    const size_t lds = (size_t)pw * tile + (size_t)sw * 2 * 3 * a.m.nR * sizeof(float);
    const dim3 g((d->B + pw - 1) / pw), blk(64 * (pw + sw));
    if (g.x == 0) return TAP_OK;
#define TAP_BT(H_, NC_, M_) do { TAP_HIP_CHECK(ctx, tap_allow_lds(k_big_transition<H_, NC_, M_>, lds)); \
        hipLaunchKernelGGL((k_big_transition<H_, NC_, M_>), g, blk, lds, st, a, pw); } while (0)
#define TAP_BT_M(H_, NC_) do { if (mode == 1) TAP_BT(H_, NC_, 1); else TAP_BT(H_, NC_, 2); } while (0)
#define TAP_BT_NC(H_) do { switch (mask_fast_path_cols(a.m)) { case 1: TAP_BT_M(H_, 1); break; case 2: TAP_BT_M(H_, 2); break; \
        default: TAP_BT_M(H_, 4); break; } } while (0)
    if (hard) TAP_BT_NC(true); else TAP_BT_NC(false);
#undef TAP_BT_NC
#undef TAP_BT_M
#undef TAP_BT
    TAP_LAUNCH_CHECK(ctx, "k_big_transition");
    return TAP_OK;
}

int tap_big_feature(tap_ctx *ctx, const tap_env_desc *d, const EnvView &v, float *out, int flen, hipStream_t st)
{
    if (d->B == 0) return TAP_OK;
    const int wpb = TAP_BLOCK / 64;
    hipLaunchKernelGGL(k_big_feature, dim3((d->B + wpb - 1) / wpb), dim3(TAP_BLOCK), 0, st, *d, v, out, flen);
    TAP_LAUNCH_CHECK(ctx, "k_big_feature");
    return TAP_OK;
}
