// macs_big.hip -- MACS / MUL 2D (tools.calc_one_position_mcs_2d, tools.py:2456-2749) for containers wider than the
// lane-per-column kernels cover (W > 64; the reference builds any --container_width, model.py:279, and its MACS has
// no size limit).  Same height-map restatement as tap_macs.h / tap_macs_wide.h (read tap_macs.h's header first:
// the reference's per-level free-interval lists are the runs of columns with hm <= z, its voxel tests are height
// comparisons, `visited` is one flag per (position, level)), written for ONE THREAD per container walking its own
// columns in global memory -- a correctness path for unusual shapes like big.hip, not a fast one.  No mask is
// involved, so neither the container nor a block has a width limit beyond the state blob's (W <= 4096).
// gfx950 only.
#include "tap_common.h"
#include "tap_place.h"

// EMS entries one step can produce: the runs of level 0 (at most (W+1)/2), one run per level z > 0 for which some
// column of the run has hm == z (a run without such a column is the same run as on the level below and is skipped,
// tools.py:2526-2528) -- charged to that column, so at most W in total -- and two per placed block (:2531-2555).
__host__ __device__ inline int macs_big_cap(int W, int n_max) { return W + (W + 1) / 2 + 2 * n_max + 2; }

// scratch ints per container in the state blob: ems[cap] as (x1 | x2 << 16, z), slots[2 cap] as (xs, Z)
size_t tap_macs_big_scratch_ints(const tap_env_desc *d) { return (size_t)6 * macs_big_cap(d->W, d->n_max); }

struct MbCtx {
    int W, H, flags;
    const int32_t *hm;
    int2 *ems, *slots;
    int cap;
};

// is the block's left edge at column c, level Z a position check_position settles on (tools.py:2571-2588):
// supported, free, and -- for hard rewards -- stable.  stab_out: is_stable_2d of that position.
__device__ static bool mb_good(const MbCtx &c, int x, int Z, int bx, bool hard, int &stab_out)
{
    stab_out = 0;
    if (x < 0 || x + bx > c.W) return false;
    int first = -1, last = -1;
    for (int k = 0; k < bx; ++k) {
        const int h = c.hm[x + k];
        if (h > Z) return false;                                             // :2576 the block's volume is not free
        if (h == Z) { if (first < 0) first = k; last = k; }
    }
    if (Z > 0 && first < 0) return false;                                    // :2574 nothing under the block
    // is_stable_2d (tools.py:839-868): the centre strictly inside (first supported, last supported + 1)
    stab_out = (Z == 0) ? 1 : ((2 * first < bx) && (2 * (bx - 1 - last) < bx));
    return stab_out || !hard;                                                // :2580-2581
}

__device__ static bool mb_taken(const MbCtx &c, int n_slots, int xs, int Z)
{
    for (int s = 0; s < n_slots; ++s)
        if (c.slots[s].x == xs && c.slots[s].y == Z) return true;
    return false;
}

// usable-space tie-break score of the candidate map (tap_macs.h: macs_adj; tools.py:2708-2736 with max_h cancelled)
__device__ static long mb_adj(const MbCtx &c, int xs, int bx, int top, int m)
{
    const int W = c.W;
    long base = 0;
    for (int j = 0; j < W; ++j) {
        const int v = (j >= xs && j < xs + bx) ? top : c.hm[j];
        bool first = true;
        int next = m, best_run = 0, run = -1;
        for (int k = 0; k < W; ++k) {
            const int hk = (k >= xs && k < xs + bx) ? top : c.hm[k];
            if (hk == v && k < j) first = false;
            if (hk > v) next = min(next, hk);
            if (hk <= v) { ++run; best_run = max(best_run, run); } else run = -1;
        }
        if (first && v < m) base += (long)(next - v) * best_run;
    }
    return base - (long)m * (W - 1);
}

__global__ void __launch_bounds__(TAP_BLOCK) k_macs2d_big_step(StepArgs a, int32_t *scratch, int cap, int lpw)
{
    const int env = tap_spread_env(lpw, a.d.B);                                  // containers spread over the waves (tap_common.h)
    const int B = a.d.B, W = a.d.W, H = a.d.H;
    if (env < 0) return;
    int32_t *hm = a.v.hm + (size_t)env * W;
    const int4 cv = reinterpret_cast<const int4 *>(a.v.cnt)[env];
    Counters cnt = {cv.x, cv.y, cv.z, cv.w};
    int bx, bz;
    if (a.static_) {
        bool badp;
        const long p = tap_col((long)a.ptr[env], a.nR, badp);
        bx = badp ? 0 : (int)a.static_[((size_t)env * a.static_rows + 1) * a.nR + p];
        bz = badp ? 0 : (int)a.static_[((size_t)env * a.static_rows + 2) * a.nR + p];
    } else if (a.blocks_dtype == TAP_DT_F32) {
        bx = (int)((const float *)a.blocks)[(size_t)env * 2];
        bz = (int)((const float *)a.blocks)[(size_t)env * 2 + 1];
    } else {
        bx = ((const int32_t *)a.blocks)[(size_t)env * 2];
        bz = ((const int32_t *)a.blocks)[(size_t)env * 2 + 1];
    }
    const bool act = !a.active || a.active[env] != 0;
    int err = 0;
    bool do_step = act;
    if (act && cnt.count >= a.d.n_max) { err |= 2; do_step = false; }
    if (act && (bx < 1 || bz < 1)) { err |= 4; do_step = false; }
    const int step = cnt.count;
    Placement res = {0, 0, 0, 0, 0};

    if (do_step) {
        MbCtx c = {W, H, a.d.flags, hm, reinterpret_cast<int2 *>(scratch + (size_t)env * 6 * cap),
                   reinterpret_cast<int2 *>(scratch + (size_t)env * 6 * cap) + cap, cap};
        const bool hard = (c.flags & TAP_F_HARD) != 0;
        const int vol = bx * bz;
        int gmax = 0;
        for (int k = 0; k < W; ++k) gmax = max(gmax, hm[k]);
        int n_ems = 0;
#define MB_PUSH(x1, z, x2)                                                                   \
    do {                                                                                     \
        if (n_ems < cap) c.ems[n_ems++] = make_int2((x1) | ((x2) << 16), (z));               \
        else err |= 16;                                                                      \
    } while (0)
        // ---- (a) per-level free runs (tools.py:2517-2529); only z = 0 and z in {hm[c]} differ from below ----------
        for (int z = 0;;) {
            if (z + bz > H) break;                                            // :2519
            int nz = INT_MAX;
            for (int k = 0; k < W;) {
                if (hm[k] > z) { ++k; continue; }
                const int x1 = k;
                bool on = false;
                while (k < W && hm[k] <= z) { on |= hm[k] == z; ++k; }        // maximal run [x1, k)
                if (x1 + bx > W) break;                                       // :2525
                if (z > 0 && !on) continue;                                   // :2526-2528 the same run below
                MB_PUSH(x1, z, k - 1);                                        // :2529
            }
            for (int k = 0; k < W; ++k)
                if (hm[k] > z) nz = min(nz, hm[k]);                           // :2520 next level that differs
            if (nz == INT_MAX) break;
            z = nz;
        }
        // ---- (b) tops of the blocks placed so far (tools.py:2531-2555); failed steps sit at (0, 0) ------------------
        for (int i = 0; i < step; ++i) {
            const int x = a.v.pos[(size_t)(i * 2) * B + env], z = a.v.pos[(size_t)(i * 2 + 1) * B + env];
            const int xx = a.v.blk[(size_t)(i * 2) * B + env], zz = a.v.blk[(size_t)(i * 2 + 1) * B + env];
            const int tz = z + zz;
            if (!(tz < H)) continue;                                          // :2535
            bool full = true;
            for (int k = x; k < x + xx && k < W; ++k) full &= hm[k] <= tz;    // the slice clips at W (:2537)
            if (full) {
                const int2 want = make_int2(x | ((x + xx - 1) << 16), tz);
                bool dup = false;                                             // :2538
                for (int k = 0; k < n_ems; ++k) dup |= c.ems[k].x == want.x && c.ems[k].y == want.y;
                if (!dup) MB_PUSH(x, tz, x + xx - 1);
            } else {
                if (x + xx - 1 >= W) { err |= 8; continue; }                  // reference: IndexError :2550
                if (hm[x] <= tz && x > 0 && hm[x - 1] <= tz) {                // :2543-2548 left part
                    int len = 0;
                    while (x + len < W && hm[x + len] <= tz) ++len;           // free columns from x rightwards
                    MB_PUSH(x, tz, x + min(len, xx) - 1);
                }
                const int xe = x + xx - 1;
                if (hm[xe] <= tz && x + xx < W && hm[x + xx] <= tz) {         // :2550-2555 right part
                    int len = 0;
                    while (xe - len >= 0 && hm[xe - len] <= tz) ++len;        // free columns from xe leftwards
                    MB_PUSH(xe - min(len, xx) + 1, tz, xe);
                }
            }
        }
        // ---- both corner walks of every EMS (tools.py:2680-2700) -> slot list ---------------------------------------
        const int X = W - bx + 1;
        int n_slots = 0;
        for (int e = 0; e < n_ems; ++e) {
            const int X1 = c.ems[e].x & 0xffff, X2 = c.ems[e].x >> 16, Z = c.ems[e].y;
            if (X1 < X) {                                                     // :2686 left corner, slide right
                for (int xs = X1; xs < X; ++xs) {
                    int st;
                    if (mb_good(c, xs, Z, bx, hard, st) && !mb_taken(c, n_slots, xs, Z)) { c.slots[n_slots++] = make_int2(xs, Z); break; }
                }
            }
            const int hi = X2 - bx + 1;                                       // :2694 right corner, slide left
            if (hi >= 0) {
                if (hi + bx > W) err |= 8;
                else
                    for (int xs = hi; xs >= 0; --xs) {
                        int st;
                        if (mb_good(c, xs, Z, bx, hard, st) && !mb_taken(c, n_slots, xs, Z)) { c.slots[n_slots++] = make_int2(xs, Z); break; }
                    }
            }
        }
        // ---- score the slots (tools.py:2590-2604) --------------------------------------------------------------------
        const int valid2 = cnt.valid + vol;
        const bool tiebreak = (c.flags & TAP_F_MCS_TIE) != 0, zero = (c.flags & TAP_F_MCS_ZERO) != 0;
        auto eval_slot = [&](int s, int &xs, int &Z, int &sum, int &stab) -> double {
            xs = c.slots[s].x; Z = c.slots[s].y;
            sum = 0;
            int first = -1, last = -1;                                        // a settled slot has max == Z
            for (int k = 0; k < bx; ++k) {
                const int h = hm[xs + k];
                sum += h;
                if (h == Z) { if (first < 0) first = k; last = k; }
            }
            stab = (Z == 0) ? 1 : ((2 * first < bx) && (2 * (bx - 1 - last) < bx));
            if (zero) return 0.0;
            int height = max(gmax, Z + bz);
            if (Z + bx > height) height = Z + bz;                             // :2594 (sic block_x)
            const int emp = cnt.empty + bx * Z - sum;                         // :2598-2599
            const double C = (double)valid2 / (double)((long long)height * W);
            const double P = (c.flags & TAP_F_USE_P) ? (double)valid2 / (double)(emp + valid2) : 0.0;
            const double S = (c.flags & TAP_F_USE_S) ? (double)(cnt.nstable + stab) / (double)(cnt.count + 1) : 0.0;
            return (C + P) + S;
        };
        double rmax = -1.0;
        int win = -1, max_height = gmax;
        for (int s = 0; s < n_slots; ++s) {
            int xs, Z, sum, stab;
            const double r = eval_slot(s, xs, Z, sum, stab);
            max_height = max(max_height, Z + bz);                             // :2719 np.max(heightmap_ems)
            if (r > rmax) { rmax = r; win = s; }                              // first maximum in list order
        }
        if (n_slots > 0 && tiebreak) {
            long best_adj = 0;
            int n_tied = 0;
            win = -1;
            for (int s = 0; s < n_slots; ++s) {
                int xs, Z, sum, stab;
                if (eval_slot(s, xs, Z, sum, stab) != rmax) continue;
                ++n_tied;
                const long adj = mb_adj(c, xs, bx, Z + bz, max(gmax, Z + bz));
                if (win < 0 || adj > best_adj) { best_adj = adj; win = s; }
            }
            const int nt = zero ? 2 * n_ems : n_tied;
            if (nt > 1 && max_height > H) err |= 1;                           // :2718 levels up to max_height
        }
        // ---- commit (tools.py:2738-2747) --------------------------------------------------------------------------------
        if (win >= 0) {
            int xs, Z, sum, stab;
            (void)eval_slot(win, xs, Z, sum, stab);
            res.placed = 1; res.x = xs; res.z = Z; res.stab = stab;
            for (int k = xs; k < xs + bx; ++k) hm[k] = Z + bz;
            cnt.valid += vol;
            cnt.empty = cnt.empty + bx * Z - sum;
            cnt.nstable += stab;
            if (Z + bz > H) err |= 1;
        }
        cnt.count += 1;
#undef MB_PUSH
    }

    if (a.feature_out) {                                                      // tools.py:3716-3744
        float *out = a.feature_out + (size_t)env * a.flen;
        if (a.d.feature == TAP_FEAT_DIFF) {
            for (int k = 0; k + 1 < W; ++k) out[k] = (float)(hm[k + 1] - hm[k]);
        } else {
            int mn = 0;
            if (a.d.feature == TAP_FEAT_ZERO) {
                mn = INT_MAX;
                for (int k = 0; k < W; ++k) mn = min(mn, hm[k]);
            }
            for (int k = 0; k < W; ++k) out[k] = (float)(hm[k] - mn);
        }
    }
    if (do_step) {
        reinterpret_cast<int4 *>(a.v.cnt)[env] = make_int4(cnt.valid, cnt.empty, cnt.nstable, cnt.count);
        a.v.pos[(size_t)(step * 2) * B + env] = res.x;
        a.v.pos[(size_t)(step * 2 + 1) * B + env] = res.z;
        a.v.stable[(size_t)step * B + env] = (uint8_t)res.stab;
        a.v.blk[(size_t)(step * 2) * B + env] = bx;                           // history the later steps read
        a.v.blk[(size_t)(step * 2 + 1) * B + env] = bz;                       // (tools.py:2531-2533), failures too
    }
    if (err) a.v.err[env] |= err;
}

int tap_macs_big_step(tap_ctx *ctx, const StepArgs &a, hipStream_t st)
{
    if (a.d.B == 0) return TAP_OK;
    if (!a.v.scratch) return tap_fail(ctx, TAP_E_INVALID, "MACS above 64 columns: the state blob has no scratch section");
    const int lpw = tap_spread_lpw(a.d.B);
    hipLaunchKernelGGL(k_macs2d_big_step, dim3(tap_spread_grid(a.d.B, lpw, TAP_BLOCK)), dim3(TAP_BLOCK), 0, st, a, a.v.scratch,
                       macs_big_cap(a.d.W, a.d.n_max), lpw);
    TAP_LAUNCH_CHECK(ctx, "k_macs2d_big_step");
    return TAP_OK;
}
