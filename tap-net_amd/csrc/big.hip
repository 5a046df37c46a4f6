// big.hip -- LB_GREEDY (tools.py:2027-2351) for containers the lane-per-cell kernels do not cover: more than
// 64 cells, or a 3D side above 8 (the reference builds W x W containers for any --container_width,
// model.py:279).  Same height-map formulation as tap_place.h, written for ONE THREAD per
// container walking its own cells -- a correctness path for unusual shapes, not a fast one; every BASELINE
// shape takes the lane-per-cell kernels.  Limits: W*L <= 4096 cells; 3D block footprints up to 8 x 8 and 2D
// blocks up to 64 wide (the support masks of the stability tests), larger blocks raise error bit 4.
#include "tap_common.h"
#include "tap_place.h"

struct BigCtx {
    int D, W, L, H, flags;
    const uint32_t *lut;
    int32_t *hm;    // this container's W*L heights (global memory)
    int32_t *keys;  // W*L ints of scratch (hard mode: candidate keys)
};

// max / support mask / sum over a footprint
__device__ static void big_scan(const BigCtx &c, int x, int y, int bx, int by, int &mx, u64 &eq, int &sum)
{
    mx = -1; eq = 0; sum = 0;
    for (int i = 0; i < bx; ++i)
        for (int j = 0; j < by; ++j) {
            const int h = c.hm[(x + i) * c.L + y + j];
            sum += h;
            const u64 bit = 1ull << (c.D == 2 ? i : i * 8 + j);
            if (h > mx) { mx = h; eq = bit; }
            else if (h == mx) eq |= bit;
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
__device__ static Placement big_place(const BigCtx &c, Counters &cnt, int &err, int bx, int by, int bz)
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
                big_scan(c, x, y, bx, by, mx, eq, sum);
                const int z = mx;
                if (z >= c.H) err |= 1;                                          // :2109 would raise IndexError
                const int stab = z == 0 ? 1 : (c.D == 2 ? tap_stable2d(bx, eq) : tap_stable3d_any(c.lut, bx, by, eq));
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
                    else if (x + bx <= W && y + by <= L) { int mx, sum; u64 eq; big_scan(c, x, y, bx, by, mx, eq, sum); k = big_key(c, x, y, mx, cls); }
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
                    big_scan(c, _x, _y, bx, by, mx, eq, sum);
                    if (z > 0 && mx < z) continue;                               // :2106 nothing underneath
                    visited[sp >> 6] |= 1ull << (sp & 63);                       // :2107
                    if (z >= c.H) { err |= 1; continue; }                        // :2109 IndexError
                    if (mx > z) continue;                                        // :2109 not free
                    const int st = z == 0 ? 1 : (c.D == 2 ? tap_stable2d(bx, eq) : tap_stable3d_any(c.lut, bx, by, eq));
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

__global__ void __launch_bounds__(TAP_BLOCK) k_big_step(StepArgs a, int32_t *scratch)
{
    const int env = blockIdx.x * TAP_BLOCK + threadIdx.x;
    const int B = a.d.B;
    if (env >= B) return;
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
    if (act && do_step && ((D == 3 && (bx > 8 || by > 8) && bx <= W && by <= L) || (D == 2 && bx > 64 && bx <= W))) {
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

__global__ void __launch_bounds__(TAP_BLOCK) k_big_feature(tap_env_desc d, EnvView v, float *out, int flen)
{
    const int env = blockIdx.x * TAP_BLOCK + threadIdx.x;
    if (env >= d.B) return;
    big_feature(d.feature, d.D, d.W, d.L, v.hm + (size_t)env * d.W * d.L, out + (size_t)env * flen);
}

int tap_big_step(tap_ctx *ctx, const StepArgs &a, void *state, hipStream_t st)
{
    const int grid = (a.d.B + TAP_BLOCK - 1) / TAP_BLOCK;
    if (grid == 0) return TAP_OK;
    (void)state;
    hipLaunchKernelGGL(k_big_step, dim3(grid), dim3(TAP_BLOCK), 0, st, a, a.v.scratch);
    TAP_LAUNCH_CHECK(ctx, "k_big_step");
    return TAP_OK;
}

int tap_big_feature(tap_ctx *ctx, const tap_env_desc *d, const EnvView &v, float *out, int flen, hipStream_t st)
{
    const int grid = (d->B + TAP_BLOCK - 1) / TAP_BLOCK;
    if (grid == 0) return TAP_OK;
    hipLaunchKernelGGL(k_big_feature, dim3(grid), dim3(TAP_BLOCK), 0, st, *d, v, out, flen);
    TAP_LAUNCH_CHECK(ctx, "k_big_feature");
    return TAP_OK;
}
