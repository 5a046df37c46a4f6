// macs.hip -- MACS / MUL 2D placement: tools.calc_one_position_mcs_2d (tools.py:2456-2749),
// re-stated on the height-map plus the placement history.
//
// The reference keeps, per container, a voxel grid and one free-interval list per level and walks
// "empty maximal spaces" (EMS) sequentially with a shared `visited` set, so a placement is serial
// per env.  Mapping: one thread per env, one wave64 per workgroup; every thread owns a private
// column of LDS ([slot][lane] layout, conflict-free) holding its height-map, its EMS list and its
// visited bitmap -- the dynamically indexed arrays that would otherwise spill to scratch.
//
// Facts used (each checked by the oracle-vs-reference differential tests and, for this kernel, by
// tests/test_gpu_parity.py against the voxel-level oracle):
//   * voxel (c, z) != 0  <=>  z < hm[c];  level_free_space[z] == maximal runs of columns with
//     hm[c] <= z  (so only z = 0 and z in {hm[c]} can open new level-EMS);
//   * the usable-space tie-break score of a candidate with height-map hm' is
//       sum_{h < max_h} maxrun_h(hm')  =  base(hm') + (max_h - max(hm')) * (W - 1),
//     and max_h is common to all tied candidates, so ties are ordered by
//     base(hm') - max(hm') * (W - 1) without knowing max_h -- selection can stream.
#include "tap_common.h"
#include "tap_place.h"

constexpr int MACS_THREADS = 64;   // one wave per workgroup
constexpr int MACS_EMS_CAP = 128;  // packed EMS entries per env
constexpr int MACS_HM_CAP = 16;    // W <= 16
constexpr int MACS_MAX_H = 256;    // visited bitmap: ceil(H / 32) words per column

__host__ __device__ inline int macs_vis_words(int H) { return (H + 31) / 32; }
__host__ __device__ inline int macs_lds_words(int W, int H) { return MACS_EMS_CAP + W * macs_vis_words(H) + MACS_HM_CAP; }

struct MacsSel {            // streaming selection state (tools.py:2708-2736)
    double best;
    int adj, x, z, stab, emp, any;
};

__global__ void __launch_bounds__(MACS_THREADS) k_macs2d_step(StepArgs a)
{
    extern __shared__ int lds[];
    const int t = threadIdx.x;
    const int env = blockIdx.x * MACS_THREADS + t;
    const int B = a.d.B, W = a.d.W, H = a.d.H, VW = macs_vis_words(a.d.H);
    if (env >= B) return; // no barriers below: every thread only touches its own LDS column
#define EMS(k) lds[(k) * MACS_THREADS + t]
#define VIS(w) lds[(MACS_EMS_CAP + (w)) * MACS_THREADS + t]
#define HM(c) lds[(MACS_EMS_CAP + W * VW + (c)) * MACS_THREADS + t]

    int gmax = 0;
    for (int c = 0; c < W; ++c) { const int h = a.v.hm[(size_t)env * W + c]; HM(c) = h; gmax = max(gmax, h); }
    const int4 c4 = reinterpret_cast<const int4 *>(a.v.cnt)[env];
    Counters cnt = {c4.x, c4.y, c4.z, c4.w};

    int bx, bz;
    if (a.static_) {
        const long p = (long)a.ptr[env];
        bx = (int)a.static_[((size_t)env * a.static_rows + 1) * a.nR + p];
        bz = (int)a.static_[((size_t)env * a.static_rows + 2) * a.nR + p];
    } else if (a.blocks_dtype == TAP_DT_F32) {
        bx = (int)((const float *)a.blocks)[(size_t)env * 2];
        bz = (int)((const float *)a.blocks)[(size_t)env * 2 + 1];
    } else {
        bx = ((const int32_t *)a.blocks)[(size_t)env * 2];
        bz = ((const int32_t *)a.blocks)[(size_t)env * 2 + 1];
    }
    const bool act = a.active ? a.active[env] != 0 : true;
    int err = 0;
    bool do_step = act;
    if (act && cnt.count >= a.d.n_max) { err |= 2; do_step = false; }
    if (act && (bx < 1 || bz < 1)) { err |= 4; do_step = false; }

    if (do_step) {
        const int step = cnt.count;
        const int hard = a.d.flags & TAP_F_HARD;
        const int vol = bx * bz;
        // the block history the later steps' EMS search reads (tools.py:2531-2533), failures too
        a.v.blk[(size_t)(step * 2) * B + env] = bx;
        a.v.blk[(size_t)(step * 2 + 1) * B + env] = bz;

        // ---- EMS list --------------------------------------------------------------------
        int n_ems = 0;
#define EMS_PUSH(x1, z, x2)                                                                  \
    do {                                                                                     \
        if (n_ems < MACS_EMS_CAP) EMS(n_ems++) = ((x1) & 0xff) | (((x2) & 0xff) << 8) | ((z) << 16); \
        else err |= 16;                                                                      \
    } while (0)
        // (a) per-level free runs (tools.py:2517-2529): only z = 0 and z in {hm[c]} differ from
        // the level below
        for (int z = 0;;) {
            if (z + bz > H) break;                                                // :2519
            int c = 0;
            while (c < W) {
                if (HM(c) > z) { ++c; continue; }
                const int x1 = c;
                bool opened = false; // some column of the run has hm == z: the run is new at z
                while (c < W && HM(c) <= z) { opened |= HM(c) == z; ++c; }
                const int x2 = c - 1;
                if (x1 + bx > W) break;                                           // :2525
                if (z > 0 && !opened) continue;                                   // :2526-2528 same run below
                EMS_PUSH(x1, z, x2);                                              // :2529
            }
            int nz = INT_MAX;                                                     // :2520 next level that differs
            for (int k = 0; k < W; ++k) if (HM(k) > z) nz = min(nz, HM(k));
            if (nz == INT_MAX) break;
            z = nz;
        }
        // (b) tops of the blocks placed so far (tools.py:2531-2555); failed steps sit at (0, 0)
        for (int i = 0; i < step; ++i) {
            const int x = a.v.pos[(size_t)(i * 2) * B + env], z = a.v.pos[(size_t)(i * 2 + 1) * B + env];
            const int xx = a.v.blk[(size_t)(i * 2) * B + env], zz = a.v.blk[(size_t)(i * 2 + 1) * B + env];
            const int tz = z + zz;
            if (!(tz < H)) continue;                                              // :2535
            bool full = true;                                                     // :2537 (slice clips at W)
            for (int c = x; c < x + xx && c < W; ++c) full &= HM(c) <= tz;
            if (full) {
                const int want = (x & 0xff) | (((x + xx - 1) & 0xff) << 8) | (tz << 16);
                bool dup = false;                                                 // :2538
                for (int k = 0; k < n_ems; ++k) dup |= EMS(k) == want;
                if (!dup) EMS_PUSH(x, tz, x + xx - 1);
            } else {
                if (x + xx - 1 >= W) { err |= 8; continue; }                      // reference: IndexError :2550
                if (HM(x) <= tz && x > 0 && HM(x - 1) <= tz) {                    // :2543-2548 left part
                    int x2 = x;
                    for (;;) {
                        if (x2 == W - 1 || HM(x2 + 1) > tz) break;
                        if (x2 == x + xx - 1) break;
                        ++x2;
                    }
                    EMS_PUSH(x, tz, x2);
                }
                if (HM(x + xx - 1) <= tz && x + xx < W && HM(x + xx) <= tz) {     // :2550-2555 right part
                    int x1 = x + xx - 1;
                    for (;;) {
                        if (x1 == 0 || HM(x1 - 1) > tz) break;
                        if (x1 == x) break;
                        --x1;
                    }
                    EMS_PUSH(x1, tz, x + xx - 1);
                }
            }
        }

        // ---- walk the two corners of every EMS with a shared visited set (:2680-2700) --------
        for (int w = 0; w < W * VW; ++w) VIS(w) = 0;
        MacsSel sel = {-1.0, INT_MIN, 0, 0, 0, 0, 0};
        const int X = W - bx + 1;
        for (int e = 0; e < n_ems; ++e) {
            const int pk = EMS(e);
            const int X1 = pk & 0xff, X2 = (pk >> 8) & 0xff, Z = pk >> 16;
            for (int side = 0; side < 2; ++side) {
                int _x, dx;
                if (side == 0) { if (!(X1 < X)) continue; _x = X1; dx = 1; }      // :2686 left corner
                else { if (!(X2 - bx + 2 > 0)) continue; _x = X2 - bx + 1; dx = -1; } // :2694 right corner
                bool ok = false;
                int sx = 0, sstab = 0, ssum = 0;
                for (; (dx > 0 ? _x < X : _x >= 0) && !ok; _x += dx) {
                    if (_x + bx > W) { err |= 8; break; }
                    const int vw = _x * VW + (Z >> 5), vb = 1 << (Z & 31);
                    if (VIS(vw) & vb) continue;                                   // :2573
                    int mx = -1, sum = 0;
                    u64 eq = 0;
                    for (int i = 0; i < bx; ++i) {
                        const int h = HM(_x + i);
                        sum += h;
                        if (h > mx) { mx = h; eq = 1ull << i; }
                        else if (h == mx) eq |= 1ull << i;
                    }
                    if (Z > 0 && mx < Z) continue;                                // :2574
                    VIS(vw) |= vb;                                                // :2575
                    if (mx > Z) continue;                                         // :2576 volume not free
                    const int st = (Z == 0) ? 1 : tap_stable2d(bx, eq);           // :2577-2585
                    if (!st && hard) continue;                                    // :2580-2581
                    ok = true; sx = _x; sstab = st; ssum = sum;
                }
                if (!ok) continue;
                // calc_C_P_S (:2590-2604)
                const int top = Z + bz;
                const int m = max(gmax, top);                    // true max of the candidate map
                int height = m;
                if (Z + bx > height) height = Z + bz;                             // :2594 (sic block_x)
                const int emp = cnt.empty + bx * Z - ssum;                        // :2598-2599
                double r = 0.0;
                if (!(a.d.flags & TAP_F_MCS_ZERO)) {                              // :2709-2712
                    const int valid2 = cnt.valid + vol;
                    const double C = (double)valid2 / (double)((long long)height * W);
                    const double P = (a.d.flags & TAP_F_USE_P) ? (double)valid2 / (double)(emp + valid2) : 0.0;
                    const double S = (a.d.flags & TAP_F_USE_S)
                                         ? (double)(cnt.nstable + sstab) / (double)(cnt.count + 1) : 0.0;
                    r = (C + P) + S;
                }
                int adj = 0;
                if (a.d.flags & TAP_F_MCS_TIE) {
                    // base = sum_{h < m} (longest free run at level h, length - 1)  (:2667-2678),
                    // evaluated piecewise between the distinct heights of the candidate map
                    auto hc = [&](int c) { return (c >= sx && c < sx + bx) ? top : HM(c); };
                    int base = 0;
                    for (int j = 0; j < W; ++j) {
                        const int v = hc(j);
                        bool first = true;
                        int next = m;
                        for (int k = 0; k < W; ++k) {
                            const int hk = hc(k);
                            if (hk == v && k < j) first = false;
                            if (hk > v) next = min(next, hk);
                        }
                        if (!first || v >= m) continue;
                        int best_run = 0, run = -1;     // run = (length - 1) of the current free run
                        for (int k = 0; k < W; ++k) {
                            if (hc(k) <= v) { ++run; best_run = max(best_run, run); }
                            else run = -1;
                        }
                        base += (next - v) * best_run;
                    }
                    adj = base - m * (W - 1);
                }
                // first settled maximum; ties resolved by the usable-space score when enabled
                if (!sel.any || r > sel.best || (r == sel.best && (a.d.flags & TAP_F_MCS_TIE) && adj > sel.adj)) {
                    sel.any = 1; sel.best = r; sel.adj = adj; sel.x = sx; sel.z = Z; sel.stab = sstab; sel.emp = emp;
                }
            }
        }

        // ---- commit (:2738-2747) ----------------------------------------------------------------
        int px = 0, pz = 0, pst = 0;
        if (sel.any) {
            px = sel.x; pz = sel.z; pst = sel.stab;
            for (int i = 0; i < bx; ++i) HM(px + i) = pz + bz;
            cnt.valid += vol;
            cnt.empty = sel.emp;
            cnt.nstable += pst;
            if (pz + bz > H) err |= 1;
        }
        cnt.count += 1;
        reinterpret_cast<int4 *>(a.v.cnt)[env] = make_int4(cnt.valid, cnt.empty, cnt.nstable, cnt.count);
        a.v.pos[(size_t)(step * 2) * B + env] = px;
        a.v.pos[(size_t)(step * 2 + 1) * B + env] = pz;
        a.v.stable[(size_t)step * B + env] = (uint8_t)pst;
        for (int c = 0; c < W; ++c) a.v.hm[(size_t)env * W + c] = HM(c);
    }
    if (err) a.v.err[env] |= err;

    if (a.feature_out) { // tools.py:3716-3744
        float *out = a.feature_out + (size_t)env * a.flen;
        if (a.d.feature == TAP_FEAT_DIFF) {
            for (int c = 0; c + 1 < W; ++c) out[c] = (float)(HM(c + 1) - HM(c));
        } else if (a.d.feature == TAP_FEAT_ZERO) {
            int mn = INT_MAX;
            for (int c = 0; c < W; ++c) mn = min(mn, HM(c));
            for (int c = 0; c < W; ++c) out[c] = (float)(HM(c) - mn);
        } else {
            for (int c = 0; c < W; ++c) out[c] = (float)HM(c);
        }
    }
#undef EMS
#undef VIS
#undef HM
#undef EMS_PUSH
}

int tap_macs2d_step(tap_ctx *ctx, const StepArgs &a, hipStream_t st)
{
    const tap_env_desc &d = a.d;
    if (d.D != 2) return tap_fail(ctx, TAP_E_UNSUPPORTED, "MACS/MUL is implemented for 2D only");
    if (d.W > MACS_HM_CAP || d.H > MACS_MAX_H)
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "MACS supports W <= %d and H <= %d", MACS_HM_CAP, MACS_MAX_H);
    if ((d.W + 1) * ((d.W + 1) / 2) + 2 * d.n_max > MACS_EMS_CAP)
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "MACS: blocks_num %d too large for the EMS list", d.n_max);
    const int grid = (d.B + MACS_THREADS - 1) / MACS_THREADS;
    if (grid == 0) return TAP_OK;
    const size_t lds = (size_t)macs_lds_words(d.W, d.H) * MACS_THREADS * sizeof(int);
    if (lds > 64 * 1024) return tap_fail(ctx, TAP_E_UNSUPPORTED, "MACS: W=%d H=%d needs %zu bytes of LDS per workgroup", d.W, d.H, lds);
    hipLaunchKernelGGL(k_macs2d_step, dim3(grid), dim3(MACS_THREADS), lds, st, a);
    TAP_LAUNCH_CHECK(ctx, "k_macs2d_step");
    return TAP_OK;
}
