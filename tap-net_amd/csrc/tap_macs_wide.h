// tap_macs_wide.h -- device code: MACS / MUL 2D placement for containers 17 .. 64 columns wide, G = 32 / 64 lanes
// per container (lane = column).  Same restatement of tools.calc_one_position_mcs_2d (tools.py:2456-2749) and
// the same three phases as tap_macs.h (read its header first); what differs is where things live: the
// height-map stays in the group's LDS slice instead of a register array per lane (a 64-entry array would not
// fit), column masks are 64-bit, and a level's `taken` mask is a 64-bit word.  Used by macs.hip's stand-alone
// step only -- the reference's own MACS runs are 5 and 7 columns wide, so this is a coverage path, not a tuned
// one (a fused transition for it was measured slower than the two launches, see transition.hip).  Since the end of
// round 4 the wave-per-container kernel of macs_big.hip is faster from 17 columns up and runs first (macs.hip);
// this one is what runs when that kernel's tile does not fit the LDS, and under TAP_MACS2D_WAVE_FROM for A/B runs.
#pragma once

#include "tap_macs.h"

// LDS words per env group: hm[G] | ems[cap] | slots[2 cap] | taken u64[H] | history (x, z, bx, bz)
__host__ __device__ constexpr int macs_wide_group_words(int G, int H, int n_max, int W)
{
    return G + 3 * macs_ems_cap(W, n_max) + 2 * H + 4 * n_max;
}

struct MacsWideLds {
    int *hm, *ems, *slots, *hist;
    u64 *taken;
    int ems_cap;
};

__device__ __forceinline__ MacsWideLds macs_wide_lds(int *base, int G, int H, int ems_cap)
{
    MacsWideLds m;
    m.hm = base;
    m.ems = base + G;
    m.ems_cap = ems_cap;                              // even, and G is even: `taken` is 8-byte aligned
    m.slots = m.ems + ems_cap;
    m.taken = reinterpret_cast<u64 *>(m.slots + 2 * ems_cap);
    m.hist = m.slots + 2 * ems_cap + 2 * H;
    return m;
}

__device__ __forceinline__ u64 mw_bits(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }   // n low bits

// column masks of the height-map: built by ballots (lane = column), identical on every lane of the group
template <int G> __device__ __forceinline__ u64 mw_ballot(bool p, int gl0)
{
    const u64 b = __ballot(p);
    return G == 64 ? b : ((b >> gl0) & 0xffffffffull);
}

// usable-space tie-break score of the candidate map (tap_macs.h macs_adj), height-map read from LDS
__device__ inline int macs_adj_wide(const int *hmv, int W, int xs, int bx, int top, int m)
{
    int base = 0;
    for (int j = 0; j < W; ++j) {
        const int v = (j >= xs && j < xs + bx) ? top : hmv[j];
        bool first = true;
        int next = m, best_run = 0, run = -1;
        for (int k = 0; k < W; ++k) {
            const int hk = (k >= xs && k < xs + bx) ? top : hmv[k];
            if (hk == v && k < j) first = false;
            if (hk > v) next = min(next, hk);
            if (hk <= v) { ++run; best_run = max(best_run, run); } else run = -1;
        }
        if (first && v < m) base += (next - v) * best_run;
    }
    return base - m * (W - 1);
}

// One placement; preconditions and results as tap_macs_place (L.taken[0..H) = 0, L.hm = the map, L.hist filled).
template <int G>
__device__ inline Placement tap_macs_place_wide(const PlaceCfg &c, const MacsWideLds &L, int cell, int gl0,
                                                int &hm, Counters &cnt, int &err, int bx, int bz, bool do_step)
{
    static_assert(G == 32 || G == 64, "wide MACS groups are 32 or 64 lanes");
    const int W = c.W, H = c.H, ems_cap = L.ems_cap;
    const bool incell = cell < W;
    Placement res = {0, 0, 0, 0, 0};
    if (!do_step) return res;
    const int hard = c.flags & TAP_F_HARD;
    const int vol = bx * bz, step = cnt.count;
    const int hx = incell ? hm : INT_MAX;                               // own column; lanes beyond W never count
    const int gmax = group_max<G>(incell ? hm : 0);
    auto mask_le = [&](int z) -> u64 { return mw_ballot<G>(hx <= z, gl0); };   // voxel (c, z) == 0
    auto mask_eq = [&](int z) -> u64 { return mw_ballot<G>(hx == z, gl0); };

    // ---- phase 1: EMS list (identical on every lane of the group) ---------------------------------
    int n_ems = 0;
#define EMSW_PUSH(x1, z, x2)                                                                 \
    do {                                                                                     \
        if (n_ems < ems_cap) L.ems[n_ems++] = ((x1) & 0xff) | (((x2) & 0xff) << 8) | ((z) << 16); \
        else err |= 16;                                                                      \
    } while (0)
    // (a) per-level free runs (tools.py:2517-2529); only z = 0 and z in {hm[c]} differ from below
    for (int z = 0;;) {
        if (z + bz > H) break;                                                // :2519
        u64 m = mask_le(z);
        const u64 on = mask_eq(z);
        while (m) {
            const int x1 = __ffsll((long long)m) - 1;
            const u64 rest = ~(m >> x1);
            const int len = rest ? __ffsll((long long)rest) - 1 : 64 - x1;    // maximal run [x1, x1+len)
            const u64 run = mw_bits(len) << x1;
            m &= ~run;
            if (x1 + bx > W) break;                                           // :2525
            if (z > 0 && !(on & run)) continue;                               // :2526-2528 same run below
            EMSW_PUSH(x1, z, x1 + len - 1);                                   // :2529
        }
        const int nz = group_min<G>(hx > z ? hx : INT_MAX);                   // :2520 next level that differs
        if (nz == INT_MAX) break;
        z = nz;
    }
    // (b) tops of the blocks placed so far (tools.py:2531-2555); failed steps sit at (0, 0)
    for (int i = 0; i < step; ++i) {
        const int x = L.hist[i * 4], z = L.hist[i * 4 + 1], xx = L.hist[i * 4 + 2], zz = L.hist[i * 4 + 3];
        const int tz = z + zz;
        if (!(tz < H)) continue;                                              // :2535
        const u64 fr = mask_le(tz);
        const u64 span = (mw_bits(xx) << x) & mw_bits(W);                     // slice clips at W (:2537)
        if ((span & ~fr) == 0) {
            const int want = (x & 0xff) | (((x + xx - 1) & 0xff) << 8) | (tz << 16);
            int dup = 0;                                                      // :2538
            for (int k = cell; k < n_ems; k += G) dup |= L.ems[k] == want;
            if (!mw_ballot<G>(dup != 0, gl0)) EMSW_PUSH(x, tz, x + xx - 1);
        } else {
            if (x + xx - 1 >= W) { err |= 8; continue; }                      // reference: IndexError :2550
            if (((fr >> x) & 1ull) && x > 0 && ((fr >> (x - 1)) & 1ull)) {    // :2543-2548 left part
                const u64 rest = ~(fr >> x);
                const int len = rest ? __ffsll((long long)rest) - 1 : 64 - x; // free columns from x rightwards
                EMSW_PUSH(x, tz, x + min(len, xx) - 1);
            }
            const int xe = x + xx - 1;
            if (((fr >> xe) & 1ull) && x + xx < W && ((fr >> (x + xx)) & 1ull)) { // :2550-2555 right part
                const u64 low = fr << (63 - xe);                              // bit xe -> bit 63
                const u64 inv = ~low;
                const int len = inv ? __clzll((long long)inv) : 64;           // free columns from xe leftwards
                EMSW_PUSH(xe - min(len, xx) + 1, tz, xe);
            }
        }
    }
    tap_wave_lds_sync();

    // ---- phase 2: both corner walks of every EMS (tools.py:2680-2700) -> slot list ------------------
    const int X = W - bx + 1;
    const u64 fpm = mw_bits(bx);
    int n_slots = 0;
    for (int e = 0; e < n_ems; ++e) {
        const int pk = L.ems[e];
        const int X1 = pk & 0xff, X2 = (pk >> 8) & 0xff, Z = pk >> 16;
        // every lane tests its own column as the block's left edge at level Z (:2571-2588)
        const u64 fr = mask_le(Z), on = mask_eq(Z);
        bool good = false;
        if (incell && cell + bx <= W) {
            const u64 eq = (on >> cell) & fpm;
            const bool free_ = ((fr >> cell) & fpm) == fpm;                   // :2576
            const bool supported = Z == 0 || eq != 0;                         // :2574
            const int stab = (Z == 0) ? 1 : (eq ? tap_stable2d(bx, eq) : 0);  // :2577-2585
            good = supported && free_ && (stab || !hard);                     // :2580-2581
        }
        const u64 gm = mw_ballot<G>(good, gl0);
        u64 tk = L.taken[Z];
        if (X1 < X) {                                                         // :2686 left corner, slide right
            const u64 m = gm & ~tk & ~mw_bits(X1);
            if (m) {
                const int xs = __ffsll((long long)m) - 1;
                tk |= 1ull << xs;
                L.slots[n_slots++] = xs | (Z << 8);
            }
        }
        const int hi = X2 - bx + 1;                                           // :2694 right corner, slide left
        if (hi >= 0) {
            if (hi + bx > W) err |= 8;
            else {
                const u64 m = gm & ~tk & mw_bits(hi + 1);
                if (m) {
                    const int xs = 63 - __clzll((long long)m);
                    tk |= 1ull << xs;
                    L.slots[n_slots++] = xs | (Z << 8);
                }
            }
        }
        L.taken[Z] = tk; // every lane stores the same value and reads back its own
    }
    tap_wave_lds_sync();

    // ---- phase 3: score the slots (tools.py:2590-2604), lanes round-robin ---------------------------
    const int valid2 = cnt.valid + vol;
    const bool tiebreak = (c.flags & TAP_F_MCS_TIE) != 0, zero = (c.flags & TAP_F_MCS_ZERO) != 0;
    auto eval_slot = [&](int s, int &xs, int &Z, int &sum, int &stab) -> double {
        const int sp = L.slots[s];
        xs = sp & 0xff; Z = sp >> 8;
        sum = 0;
        u64 eq = 0;                                                           // a settled slot has max == Z
        for (int k = 0; k < bx; ++k) {
            const int h = L.hm[xs + k];
            sum += h;
            eq |= (u64)(h == Z) << k;
        }
        stab = (Z == 0) ? 1 : tap_stable2d(bx, eq);
        if (zero) return 0.0;
        int height = max(gmax, Z + bz);
        if (Z + bx > height) height = Z + bz;                                 // :2594 (sic block_x)
        const int emp = cnt.empty + bx * Z - sum;                             // :2598-2599
        const double C = (double)valid2 / (double)((long long)height * W);
        const double P = (c.flags & TAP_F_USE_P) ? (double)valid2 / (double)(emp + valid2) : 0.0;
        const double S = (c.flags & TAP_F_USE_S) ? (double)(cnt.nstable + stab) / (double)(cnt.count + 1) : 0.0;
        return (C + P) + S;
    };
    double my_r = -1.0;
    int my_slot = INT_MAX; // order index of this lane's best slot
    for (int s = cell; s < n_slots; s += G) {
        int xs, Z, sum, stab;
        const double r = eval_slot(s, xs, Z, sum, stab);
        if (r > my_r) { my_r = r; my_slot = s; } // slots come in increasing order: first maximum kept
    }
    const double rmax = group_fmax<G>(my_r);
    int win = INT_MAX;
    if (n_slots > 0) {
        if (!tiebreak) {
            win = group_min<G>(my_r == rmax ? my_slot : INT_MAX);
        } else {
            int best_adj = INT_MIN, best_s = INT_MAX, n_tied = 0, max_height = gmax;
            for (int s = cell; s < n_slots; s += G) {
                int xs, Z, sum, stab;
                const double r = eval_slot(s, xs, Z, sum, stab);
                max_height = max(max_height, Z + bz);                         // :2719 np.max(heightmap_ems)
                if (r != rmax) continue;
                ++n_tied;
                const int adj = macs_adj_wide(L.hm, W, xs, bx, Z + bz, max(gmax, Z + bz));
                if (adj > best_adj) { best_adj = adj; best_s = s; }
            }
            group_butterfly<G>((int)(threadIdx.x & 63), [&](auto get) { // lexicographic (adj desc, order asc)
                const int a2 = get(best_adj), s2 = get(best_s);
                if (a2 > best_adj || (a2 == best_adj && s2 < best_s)) { best_adj = a2; best_s = s2; }
            });
            win = best_s;
            const int nt = zero ? 2 * n_ems : group_sum<G>(n_tied);
            if (nt > 1 && group_max<G>(max_height) > H) err |= 1;              // :2718 levels up to max_height
        }
    }

    // ---- commit (tools.py:2738-2747) -------------------------------------------------------------------
    if (win != INT_MAX) {
        int xs, Z, sum, stab;
        (void)eval_slot(win, xs, Z, sum, stab);
        res.placed = 1; res.x = xs; res.z = Z; res.stab = stab;
        if (incell && cell >= xs && cell < xs + bx) hm = Z + bz;
        cnt.valid += vol;
        cnt.empty = cnt.empty + bx * Z - sum;
        cnt.nstable += stab;
        if (Z + bz > H) err |= 1;
    }
    cnt.count += 1;
#undef EMSW_PUSH
    return res;
}

// One env's whole step as executed by its lane group (macs.hip's stand-alone step): load state + history, place,
// store state and feature; flags: TAP_T_FRESH / TAP_T_RATIO as in tap_macs3_wave (fresh-container start, calc_ratio).
template <int G>
__device__ inline void tap_macs_wide_wave(const StepArgs &a, int flags, float *ratio_out, int env, int cell, int lane,
                                          int *lds_group)
{
    const int B = a.d.B, W = a.d.W, H = a.d.H;
    const bool ev = env < B, incell = cell < W, fresh = flags & TAP_T_FRESH;
    const int gl0 = lane - cell;
    const MacsWideLds L = macs_wide_lds(lds_group, G, H, macs_ems_cap(W, a.d.n_max));

    int hm = (ev && incell && !fresh) ? a.v.hm[(size_t)env * W + cell] : 0;
    const int cv = (ev && cell < 4 && !fresh) ? a.v.cnt[(size_t)env * 4 + cell] : 0;
    Counters cnt = {__shfl(cv, gl0), __shfl(cv, gl0 + 1), __shfl(cv, gl0 + 2), __shfl(cv, gl0 + 3)};
    int bx = 1, bz = 1;
    bool act = ev;
    if (ev) {
        if (a.static_) {
            bool badp;
            const long p = tap_col((long)a.ptr[env], a.nR, badp);
            const float vx = a.static_[((size_t)env * a.static_rows + 1) * a.nR + p];
            const float vz = a.static_[((size_t)env * a.static_rows + 2) * a.nR + p];
            bx = badp ? 0 : (int)vx;
            bz = badp ? 0 : (int)vz;
        } else if (a.blocks_dtype == TAP_DT_F32) {
            bx = (int)((const float *)a.blocks)[(size_t)env * 2];
            bz = (int)((const float *)a.blocks)[(size_t)env * 2 + 1];
        } else {
            bx = ((const int32_t *)a.blocks)[(size_t)env * 2];
            bz = ((const int32_t *)a.blocks)[(size_t)env * 2 + 1];
        }
        if (a.active) act = a.active[env] != 0;
    }
    int err = 0;
    bool do_step = act;
    if (act && cnt.count >= a.d.n_max) { err |= 2; do_step = false; }
    if (act && (bx < 1 || bz < 1)) { err |= 4; do_step = false; }

    L.hm[cell] = hm;
    for (int i = cell; i < H; i += G) L.taken[i] = 0;
    if (ev) // one round trip for the whole placement history
        for (int k = cell; k < cnt.count * 4 && k < a.d.n_max * 4; k += G) {
            const int i = k >> 2, f = k & 3;
            L.hist[k] = (f < 2 ? a.v.pos : a.v.blk)[(size_t)(i * 2 + (f & 1)) * B + env];
        }
    tap_wave_lds_sync();
    const int step = cnt.count;
    const PlaceCfg cfg = {W, 1, H, a.d.flags, nullptr};
    // an env beyond the batch shares its wave with live ones: its group still runs the (empty) placement so
    // that every ballot of the wave is executed by all of its lanes
    const Placement pl = tap_macs_place_wide<G>(cfg, L, cell, gl0, hm, cnt, err, bx, bz, do_step);
    err = group_or<G>(err);

    tap_wave_lds_sync();
    L.hm[cell] = hm;
    tap_wave_lds_sync();
    const int gmax = (flags & TAP_T_RATIO) ? group_max<G>(incell ? hm : 0) : 0;
    if (ev) {
        if (incell && (do_step || fresh)) a.v.hm[(size_t)env * W + cell] = hm;
        if (a.feature_out)
            tap_write_feature<2, G>(a.d.feature, W, 1, L.hm, cell, hm, a.feature_out + (size_t)env * a.flen);
        if (cell == 0) {
            if (do_step || fresh)
                reinterpret_cast<int4 *>(a.v.cnt)[env] = make_int4(cnt.valid, cnt.empty, cnt.nstable, cnt.count);
            if (do_step) {
                a.v.pos[(size_t)(step * 2) * B + env] = pl.x;
                a.v.pos[(size_t)(step * 2 + 1) * B + env] = pl.z;
                a.v.stable[(size_t)step * B + env] = (uint8_t)pl.stab;
                a.v.blk[(size_t)(step * 2) * B + env] = bx;   // history the later steps read
                a.v.blk[(size_t)(step * 2 + 1) * B + env] = bz; // (tools.py:2531-2533), failures too
            }
            if (fresh) a.v.err[env] = err;
            else if (err) a.v.err[env] |= err;
            if (flags & TAP_T_RATIO) {                         // tools.py:3887-3966
                double C = 0.0, P = 0.0, S = 0.0;
                if (cnt.count != 0) {
                    C = (double)cnt.valid / (double)((long long)W * gmax);
                    P = (double)cnt.valid / (double)(cnt.empty + cnt.valid);
                    S = (double)cnt.nstable / (double)cnt.count;
                }
                ratio_out[env] = (float)tap_ratio_formula(a.d.ratio_mode, C, P, S);
            }
        }
    } else if (a.d.feature == TAP_FEAT_ZERO) {
        (void)group_min<G>(INT_MAX);
    }
}
