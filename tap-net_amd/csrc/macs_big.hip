// macs_big.hip -- MACS / MUL 2D (tools.calc_one_position_mcs_2d, tools.py:2456-2749) for containers wider than the
// lane-per-column kernels handle well (W > 32; the reference builds any --container_width, model.py:279, and its MACS
// has no size limit).  Same height-map restatement as tap_macs.h / tap_macs_wide.h (read tap_macs.h's header first:
// the reference's per-level free-interval lists are the runs of columns with hm <= z, its voxel tests are height
// comparisons, `visited` is one flag per (position, level)).  Two forms: k_macs2d_wave_step, ONE WAVEFRONT per container
// with the long loops shared by the lanes (the form that runs), and k_macs2d_big_step, one THREAD per container walking
// its own columns in global memory (the serial statement of the algorithm; the fallback when a container's tile does not
// fit the LDS).  No mask is involved, so neither the container nor a block has a width limit beyond the state
// blob's (W <= 4096).  gfx950 only.
#include "tap_common.h"
#include "tap_place.h"
#include "tap_masks.h"
#include "tap_transition.h"

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

// ---- one WAVEFRONT per container (round 4) ------------------------------------------------------------------------------
// The same algorithm as k_macs2d_big_step, statement for statement, with its control skeleton run wave-uniformly on the
// container's LDS tile and the long loops shared by the lanes: a level's column runs come out of 64-column ballots, "is
// the block's top free" / "how far is it free" are wave reductions, the settling level and stability of every position are
// computed one position per lane once per step (a position settles at Z iff the maximum under it IS Z, so `visited`
// is one flag per position), a corner walk is a wave-wide min / max over the settling positions of its range, the slots
// are scored one per lane and the usable-space score of a tied slot sums one column per lane.
// Tile (ints): hm[W] | lev[W] | psum[W] | slots[W] | tl[W + 1] | tr[W + 1] | ems[2 cap] | history pos[2 n_max] | blk[2 n_max]
__host__ __device__ inline size_t macs_wave_tile_ints(int W, int cap, int n_max) { return (size_t)6 * W + 2 + (size_t)2 * cap + (size_t)4 * n_max; }

__device__ __forceinline__ int mw_min(int v) { return group_min<64>(v); }     // DPP + readlane (tap_place.h): all 64 lanes call
__device__ __forceinline__ int mw_max(int v) { return group_max<64>(v); }
__device__ __forceinline__ long mw_sum(long v)
{
    for (int o = 32; o > 0; o >>= 1) v += ((long)__shfl_xor((int)(v >> 32), o) << 32) | (unsigned)__shfl_xor((int)v, o);
    return v;
}

// one MACS 2D step of container `env` by one wavefront (every lane calls; env < B); tile = the wave's LDS tile
__device__ __forceinline__ void macs2d_wave_body(const StepArgs &a, int cap, int env, int lane, int32_t *tile)
{
    const int B = a.d.B, W = a.d.W, H = a.d.H;
    int32_t *hm = tile;
    int32_t *lev = hm + W, *psum = lev + W, *slots = psum + W;
    int32_t *tl = slots + W, *tr = tl + W + 1;                                  // the tie-break's per-level tables
    int2 *ems = reinterpret_cast<int2 *>(tr + W + 1);
    int32_t *ghm = a.v.hm + (size_t)env * W;
    int gmax = 0;
    for (int k = lane; k < W; k += 64) { const int h = ghm[k]; hm[k] = h; gmax = max(gmax, h); }
    gmax = mw_max(gmax);
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
    int32_t *hpos = reinterpret_cast<int32_t *>(ems + cap), *hblk = hpos + 2 * a.d.n_max;   // the history so far, one round trip for all of it
    for (int k = lane; k < 2 * min(step, a.d.n_max); k += 64) { hpos[k] = a.v.pos[(size_t)k * B + env]; hblk[k] = a.v.blk[(size_t)k * B + env]; }
    tap_wave_lds_sync();

    if (do_step) {                                                               // wave-uniform
        const int flags = a.d.flags;
        const bool hard = (flags & TAP_F_HARD) != 0;
        const int vol = bx * bz;
        int n_ems = 0;
#define MW_PUSH(x1, z, x2)                                                                   \
    do {                                                                                     \
        if (n_ems < cap) { if (lane == 0) ems[n_ems] = make_int2((x1) | ((x2) << 16), (z)); ++n_ems; } \
        else err |= 16;                                                                      \
    } while (0)
        // ---- (a) per-level free runs (tools.py:2517-2529); only z = 0 and z in {hm[c]} differ from below ----------
        for (int z = 0;;) {
            if (z + bz > H) break;                                                // :2519
            bool in_run = false, on = false, stop = false;
            int x1 = 0;
            auto close_run = [&](int x2) {                                        // the run [x1, x2] is complete
                if (x1 + bx > W) { stop = true; return; }                         // :2525
                if (z > 0 && !on) return;                                         // :2526-2528 the same run below
                MW_PUSH(x1, z, x2);                                               // :2529
            };
            for (int c0 = 0; c0 < W && !stop; c0 += 64) {
                const int k = c0 + lane;
                const int h = k < W ? hm[k] : INT_MAX;
                const u64 le = __ballot(h <= z), eq = __ballot(h == z);
                int pos = 0;
                while (pos < 64 && !stop) {
                    if (in_run) {
                        const u64 t = le >> pos;
                        const int len = t == ~0ull ? 64 : __ffsll((long long)~t) - 1;   // consecutive ones from pos (t has zeros shifted in unless pos = 0)
                        if (len > 0) on = on || ((eq >> pos) & (len >= 64 ? ~0ull : ((1ull << len) - 1ull))) != 0ull;
                        pos += len;
                        if (pos < 64) { in_run = false; close_run(c0 + pos - 1); }
                    } else {
                        const u64 t = le >> pos;
                        if (t == 0ull) { pos = 64; break; }
                        pos += __ffsll((long long)t) - 1;
                        in_run = true; x1 = c0 + pos; on = false;
                    }
                }
            }
            if (in_run && !stop) close_run(W - 1);
            int nz = INT_MAX;
            for (int k = lane; k < W; k += 64) { const int h = hm[k]; if (h > z) nz = min(nz, h); }   // :2520 next level that differs
            nz = mw_min(nz);
            if (nz == INT_MAX) break;
            z = nz;
        }
        // ---- (b) tops of the blocks placed so far (tools.py:2531-2555); failed steps sit at (0, 0) ------------------
        for (int i = 0; i < step; ++i) {
            const int x = hpos[i * 2], z = hpos[i * 2 + 1];
            const int xx = hblk[i * 2], zz = hblk[i * 2 + 1];
            const int tz = z + zz;
            if (!(tz < H)) continue;                                              // :2535
            bool covered = false;                                                 // the slice clips at W (:2537)
            for (int k = x + lane; k < x + xx && k < W; k += 64) covered |= hm[k] > tz;
            const bool full = __ballot(covered) == 0ull;
            if (full) {
                const int2 want = make_int2(x | ((x + xx - 1) << 16), tz);
                tap_wave_lds_sync();
                bool dup = false;                                                 // :2538
                for (int k = lane; k < n_ems; k += 64) dup |= ems[k].x == want.x && ems[k].y == want.y;
                if (__ballot(dup) == 0ull) MW_PUSH(x, tz, x + xx - 1);
            } else {
                if (x + xx - 1 >= W) { err |= 8; continue; }                      // reference: IndexError :2550
                const int xe = x + xx - 1;
                int first_hi = W, last_hi = -1;                                   // first column >= x / last column <= xe above tz
                for (int k = lane; k < W; k += 64) {
                    const bool hi = hm[k] > tz;
                    if (hi && k >= x) first_hi = min(first_hi, k);
                    if (hi && k <= xe) last_hi = max(last_hi, k);
                }
                first_hi = mw_min(first_hi); last_hi = mw_max(last_hi);
                if (hm[x] <= tz && x > 0 && hm[x - 1] <= tz)                      // :2543-2548 left part
                    MW_PUSH(x, tz, x + min(first_hi - x, xx) - 1);
                if (hm[xe] <= tz && x + xx < W && hm[x + xx] <= tz)               // :2550-2555 right part
                    MW_PUSH(xe - min(xe - last_hi, xx) + 1, tz, xe);
            }
        }
        // ---- per position: settling level, stability, height sum (mb_good) ------------------------------------------
        const int X = W - bx + 1;
        for (int xs = lane; xs < W; xs += 64) {
            int v = -1, sum = 0;
            if (xs < X) {
                int mx = -1, first = -1, last = -1;
                for (int k = 0; k < bx; ++k) {
                    const int h = hm[xs + k];
                    sum += h;
                    if (h > mx) { mx = h; first = last = k; }
                    else if (h == mx) last = k;
                }
                // is_stable_2d (tools.py:839-868): the centre strictly inside (first supported, last supported + 1)
                const int stab = (mx == 0) ? 1 : ((2 * first < bx) && (2 * (bx - 1 - last) < bx));
                if (stab || !hard) v = (mx << 2) | (stab << 1);                  // :2580-2581
            }
            lev[xs] = v; psum[xs] = sum;
        }
        tap_wave_lds_sync();
        // ---- both corner walks of every EMS (tools.py:2680-2700) -> slot list ---------------------------------------
        int n_slots = 0;
        auto settle = [&](int xs) {
            if (lane == 0) { lev[xs] |= 1; slots[n_slots] = xs; }
            ++n_slots;
            tap_wave_lds_sync();
        };
        for (int e = 0; e < n_ems; ++e) {
            const int X1 = ems[e].x & 0xffff, X2 = ems[e].x >> 16, Z = ems[e].y;
            if (X1 < X) {                                                         // :2686 left corner, slide right
                int best = INT_MAX;
                for (int xs = X1 + lane; xs < X; xs += 64) { const int v = lev[xs]; if (v >= 0 && !(v & 1) && (v >> 2) == Z) best = min(best, xs); }
                best = mw_min(best);
                if (best != INT_MAX) settle(best);
            }
            const int hi = X2 - bx + 1;                                           // :2694 right corner, slide left
            if (hi >= 0) {
                if (hi + bx > W) err |= 8;
                else {
                    int best = -1;
                    for (int xs = lane; xs <= hi; xs += 64) { const int v = lev[xs]; if (v >= 0 && !(v & 1) && (v >> 2) == Z) best = max(best, xs); }
                    best = mw_max(best);
                    if (best >= 0) settle(best);
                }
            }
        }
        // ---- score the slots (tools.py:2590-2604), one per lane -------------------------------------------------------
        const int valid2 = cnt.valid + vol;
        const bool tiebreak = (flags & TAP_F_MCS_TIE) != 0, zero = (flags & TAP_F_MCS_ZERO) != 0;
        auto eval_slot = [&](int s, int &xs, int &Z, int &sum, int &stab) -> double {
            xs = slots[s] & 0xffff;
            const int v = lev[xs];
            Z = v >> 2; stab = (v >> 1) & 1; sum = psum[xs];
            if (zero) return 0.0;
            int height = max(gmax, Z + bz);
            if (Z + bx > height) height = Z + bz;                                 // :2594 (sic block_x)
            const int emp = cnt.empty + bx * Z - sum;                             // :2598-2599
            const double C = (double)valid2 / (double)((long long)height * W);
            const double P = (flags & TAP_F_USE_P) ? (double)valid2 / (double)(emp + valid2) : 0.0;
            const double S = (flags & TAP_F_USE_S) ? (double)(cnt.nstable + stab) / (double)(cnt.count + 1) : 0.0;
            return (C + P) + S;
        };
        double rmax = -1.0;
        int win = INT_MAX, max_height = gmax;
        for (int s = lane; s < n_slots; s += 64) {
            int xs, Z, sum, stab;
            const double r = eval_slot(s, xs, Z, sum, stab);
            max_height = max(max_height, Z + bz);                                 // :2719 np.max(heightmap_ems)
            if (r > rmax) { rmax = r; win = s; }                                  // first maximum in list order
        }
        for (int o = 32; o > 0; o >>= 1) {
            const double r2 = __hiloint2double(__shfl_xor(__double2hiint(rmax), o), __shfl_xor(__double2loint(rmax), o));
            const int w2 = __shfl_xor(win, o);
            if (r2 > rmax || (r2 == rmax && w2 < win)) { rmax = r2; win = w2; }
        }
        max_height = mw_max(max_height);
        if (n_slots == 0) win = -1;
        if (n_slots > 0 && tiebreak) {
            int n_tied = 0;
            for (int s0 = 0; s0 < n_slots; s0 += 64) {
                const int sl = s0 + lane;
                bool tie = false;
                if (sl < n_slots) { int xs, Z, sum, stab; tie = eval_slot(sl, xs, Z, sum, stab) == rmax; }
                if (tie) slots[sl] |= 0x10000;                                    // marks the tie-break's candidates
                n_tied += __popcll(__ballot(tie));
            }
            tap_wave_lds_sync();
            // usable-space score of a candidate map (mb_adj) = sum over the levels h from its lowest column up to
            // m = max(gmax, top) of (longest run of columns <= h) - 1, minus m (W - 1).  Below the block's top its
            // footprint is filled, so the longest run lies left or right of it; from the top up the map is the old one.
            // Per level of the OLD map ONE pair of tables -- tl[a] = longest free run within columns [0, a), tr[a] =
            // within [a, W) -- serves every candidate; the candidates sit one per lane.
            if (n_tied > 1) {
                long best_adj = 0;
                win = -1;
                int hmin = INT_MAX;
                for (int k = lane; k < W; k += 64) hmin = min(hmin, hm[k]);
                hmin = mw_min(hmin);
                const int nch = (W + 63) / 64;
                for (int s0 = 0; s0 < n_slots; s0 += 64) {                        // candidates in list order, 64 at a time
                    const int sl = s0 + lane;
                    const int q = sl < n_slots ? slots[sl] : 0;
                    const bool mine = (q & 0x10000) != 0;
                    if (__ballot(mine) == 0ull) continue;
                    const int xs = q & 0xffff;
                    const int top = (mine ? lev[xs] >> 2 : 0) + bz, m = max(gmax, top);
                    const int mtop = mw_max(mine ? m : 0);
                    long base = 0;
                    for (int lo = hmin;;) {
                        int nxt = INT_MAX;
                        for (int k = lane; k < W; k += 64) { const int h = hm[k]; if (h > lo && h < nxt) nxt = h; }
                        nxt = mw_min(nxt);
                        int ce = 0, cm = 0;                                       // run ending at the previous chunk's last column; best so far
                        if (lane == 0) { tl[0] = 0; tr[W] = 0; }
                        for (int ch = 0; ch < nch; ++ch) {
                            const int k = ch * 64 + lane;
                            const u64 le = __ballot(k < W && hm[min(k, W - 1)] <= lo);
                            const u64 t = ~le << (63 - lane);
                            int e = t == 0ull ? lane + 1 + ce : __clzll((long long)t);    // free columns ending at k
                            const int e63 = __shfl(e, 63);
                            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(e, o); if (lane >= o) e = max(e, u); }
                            e = max(e, cm);
                            if (k < W) tl[k + 1] = e;
                            ce = e63; cm = __shfl(e, 63);
                        }
                        ce = 0; cm = 0;
                        for (int ch = nch - 1; ch >= 0; --ch) {
                            const int k = ch * 64 + lane;
                            const u64 le = __ballot(k < W && hm[min(k, W - 1)] <= lo);
                            const int c = __ffsll((long long)~(le >> lane)) - 1;  // free columns starting at k (zeros are shifted in: c <= 64 - lane)
                            int e = (lane == 0 && le == ~0ull) ? 64 + ce : (c == 64 - lane ? c + ce : c);
                            const int e0 = __shfl(e, 0);
                            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_down(e, o); if (lane + o < 64) e = max(e, u); }
                            e = max(e, cm);
                            if (k < W) tr[k] = e;
                            ce = e0; cm = __shfl(e, 0);
                        }
                        tap_wave_lds_sync();
                        if (mine && lo < m) {
                            const int hi = min(nxt, m), a_hi = min(hi, top), b_lo = max(lo, top);
                            const int side = max(tl[xs], tr[xs + bx]);
                            if (a_hi > lo && side >= 1) base += (long)(a_hi - lo) * (side - 1);
                            if (hi > b_lo) base += (long)(hi - b_lo) * (tl[W] - 1);
                        }
                        tap_wave_lds_sync();
                        if (nxt == INT_MAX || nxt >= mtop) break;
                        lo = nxt;
                    }
                    long adj = mine ? base - (long)m * (W - 1) : LONG_MIN;
                    int wsl = mine ? sl : INT_MAX;
                    for (int o = 32; o > 0; o >>= 1) {
                        const long a2 = ((long)__shfl_xor((int)(adj >> 32), o) << 32) | (unsigned)__shfl_xor((int)adj, o);
                        const int w2 = __shfl_xor(wsl, o);
                        if (a2 > adj || (a2 == adj && w2 < wsl)) { adj = a2; wsl = w2; }
                    }
                    if (win < 0 || adj > best_adj) { best_adj = adj; win = wsl; }
                }
            }
            const int nt = zero ? 2 * n_ems : n_tied;
            if (nt > 1 && max_height > H) err |= 1;                               // :2718 levels up to max_height
        }
        // ---- commit (tools.py:2738-2747) --------------------------------------------------------------------------------
        if (win >= 0) {
            int xs, Z, sum, stab;
            (void)eval_slot(win, xs, Z, sum, stab);
            res.placed = 1; res.x = xs; res.z = Z; res.stab = stab;
            tap_wave_lds_sync();
            for (int k = xs + lane; k < xs + bx; k += 64) { hm[k] = Z + bz; ghm[k] = Z + bz; }
            cnt.valid += vol;
            cnt.empty = cnt.empty + bx * Z - sum;
            cnt.nstable += stab;
            if (Z + bz > H) err |= 1;
        }
        cnt.count += 1;
        tap_wave_lds_sync();
#undef MW_PUSH
    }

    if (a.feature_out) {                                                          // tools.py:3716-3744
        float *out = a.feature_out + (size_t)env * a.flen;
        if (a.d.feature == TAP_FEAT_DIFF) {
            for (int k = lane; k + 1 < W; k += 64) out[k] = (float)(hm[k + 1] - hm[k]);
        } else {
            int mn = 0;
            if (a.d.feature == TAP_FEAT_ZERO) {
                mn = INT_MAX;
                for (int k = lane; k < W; k += 64) mn = min(mn, hm[k]);
                mn = mw_min(mn);
            }
            for (int k = lane; k < W; k += 64) out[k] = (float)(hm[k] - mn);
        }
    }
    if (lane == 0) {
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
}

__global__ void __launch_bounds__(TAP_BLOCK) k_macs2d_wave_step(StepArgs a, int cap)
{
    extern __shared__ int32_t mw_lds[];
    const int lane = threadIdx.x & 63, wave_in_wg = TAP_WAVE_INDEX();
    const int env = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (env >= a.d.B) return;                                                     // wave-uniform
    macs2d_wave_body(a, cap, env, lane, mw_lds + (size_t)wave_in_wg * macs_wave_tile_ints(a.d.W, cap, a.d.n_max));
}

// The decoding step in ONE launch (round 5): a container's wavefront runs update_dynamic + update_mask of its own
// precedence slab on the bit shadow (tap_transition.h) and then its placement -- round 4 ran a mask launch and a
// placement launch.  MODE 1: on the shadow, 2: the episode's first step.
template <int NC, int MODE>
__global__ void __launch_bounds__(TAP_BLOCK) k_macs2d_wave_transition(TransArgs a, int cap, int PW, int tile_ints)
{
    extern __shared__ int32_t mw_lds[];
    const int wave = TAP_WAVE_INDEX(), lane = threadIdx.x & 63;
    const int env = blockIdx.x * PW + wave;
    if (env >= a.s.d.B) return;                                                   // wave-uniform
    // The wave first runs its container's precedence update (one slab: inputs in one round trip, write-through stores
    // that drain while the placement runs), then the placement.  Stream waves of their own, as in k_transition, would
    // occupy wave slots at this kernel's register count: with 4 + 2 waves per workgroup a CU held 8 placement waves
    // instead of 16 and the step took 218 against 148 us (MACS 3D 10 x 10, B = 4 096, round 5).
    trans_stream_wave<1, NC, MODE>(a.m, env, lane, reinterpret_cast<float *>(mw_lds + (size_t)PW * tile_ints) + (size_t)wave * 3 * a.m.nR);
    if (lane == 0 && a.s.static_ && (a.s.dec_static_out || a.s.tour_out || a.s.picked_out)) {   // the gather's by-products
        bool badp;
        const long praw = (long)a.s.ptr[env];
        const long p = tap_col(praw, a.s.nR, badp);
        float fv[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < a.s.d.D; ++k) fv[k] = badp ? 0.f : a.s.static_[((size_t)env * a.s.static_rows + 1 + k) * a.s.nR + p];
        tap_step_aux(a.s, env, a.s.d.D, fv, praw);
    }
    macs2d_wave_body(a.s, cap, env, lane, mw_lds + (size_t)wave * tile_ints);
}

static int macs2d_transition_pw(const tap_ctx *ctx, const tap_env_desc *d, int nR)
{
    if (tap_wave_kernels_off()) return 0;
    const size_t tile = macs_wave_tile_ints(d->W, macs_big_cap(d->W, d->n_max), d->n_max) * sizeof(int32_t);
    for (int pw = 4; pw >= 1; pw >>= 1)
        if ((size_t)pw * tile + (size_t)pw * 3 * nR * sizeof(float) <= tap_lds_limit(ctx)) return pw;
    return 0;
}

bool tap_macs_wave_transition_ok(const tap_ctx *ctx, const tap_env_desc *d, int nR) { return macs2d_transition_pw(ctx, d, nR) > 0; }

// mask update + placement of a MACS 2D container above 16 columns in one launch (the caller resets / emits calc_ratio)
int tap_macs_wave_transition(tap_ctx *ctx, const tap_env_desc *d, const TransArgs &a, hipStream_t st)
{
    const int pw = macs2d_transition_pw(ctx, d, a.m.nR);
    if (pw == 0) return tap_fail(ctx, TAP_E_UNSUPPORTED, "no fused step for this container");
    const int cap = macs_big_cap(d->W, d->n_max), tile_ints = (int)macs_wave_tile_ints(d->W, cap, d->n_max);
    const int mode = a.m.bits_in ? 1 : 2;
    const size_t lds = (size_t)pw * tile_ints * sizeof(int32_t) + (size_t)pw * 3 * a.m.nR * sizeof(float);
    const dim3 g((d->B + pw - 1) / pw), blk(64 * pw);
    if (g.x == 0) return TAP_OK;
#define TAP_MT(NC_, M_) do { TAP_HIP_CHECK(ctx, tap_allow_lds(k_macs2d_wave_transition<NC_, M_>, lds)); \
        hipLaunchKernelGGL((k_macs2d_wave_transition<NC_, M_>), g, blk, lds, st, a, cap, pw, tile_ints); } while (0)
#define TAP_MT_M(NC_) do { if (mode == 1) TAP_MT(NC_, 1); else TAP_MT(NC_, 2); } while (0)
    switch (mask_fast_path_cols(a.m)) { case 1: TAP_MT_M(1); break; case 2: TAP_MT_M(2); break; default: TAP_MT_M(4); break; }
#undef TAP_MT_M
#undef TAP_MT
    TAP_LAUNCH_CHECK(ctx, "k_macs2d_wave_transition");
    return TAP_OK;
}

// -> TAP_OK when launched, TAP_E_UNSUPPORTED (no message) when a container's tile does not fit a wave's share of the LDS
int tap_macs_wave_step(tap_ctx *ctx, const StepArgs &a, hipStream_t st)
{
    if (a.d.B == 0) return TAP_OK;
    const int cap = macs_big_cap(a.d.W, a.d.n_max);
    const size_t tile = macs_wave_tile_ints(a.d.W, cap, a.d.n_max) * sizeof(int32_t);
    int waves = TAP_BLOCK / 64;
    while (waves > 1 && (size_t)waves * tile > tap_lds_limit(ctx)) waves >>= 1;
    if ((size_t)waves * tile > tap_lds_limit(ctx) || tap_wave_kernels_off()) return TAP_E_UNSUPPORTED;
    TAP_HIP_CHECK(ctx, tap_allow_lds(k_macs2d_wave_step, (size_t)waves * tile));
    hipLaunchKernelGGL(k_macs2d_wave_step, dim3((a.d.B + waves - 1) / waves), dim3(waves * 64), (size_t)waves * tile, st, a, cap);
    TAP_LAUNCH_CHECK(ctx, "k_macs2d_wave_step");
    return TAP_OK;
}

int tap_macs_big_step(tap_ctx *ctx, const StepArgs &a, hipStream_t st)
{
    if (a.d.B == 0) return TAP_OK;
    const int rc_w = tap_macs_wave_step(ctx, a, st);
    if (rc_w != TAP_E_UNSUPPORTED) return rc_w;                                  // launched, or a real error: only "does not fit" falls through
    if (!a.v.scratch) return tap_fail(ctx, TAP_E_INVALID, "MACS above 64 columns: the state blob has no scratch section");           // one wavefront per container when its tile fits the LDS
    const int lpw = tap_spread_lpw(a.d.B);
    hipLaunchKernelGGL(k_macs2d_big_step, dim3(tap_spread_grid(a.d.B, lpw, TAP_BLOCK)), dim3(TAP_BLOCK), 0, st, a, a.v.scratch,
                       macs_big_cap(a.d.W, a.d.n_max), lpw);
    TAP_LAUNCH_CHECK(ctx, "k_macs2d_big_step");
    return TAP_OK;
}
