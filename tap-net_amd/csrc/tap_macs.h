// tap_macs.h -- device code: one MACS / MUL 2D placement for one container, G lanes per container
// (lane = container column).  tools.calc_one_position_mcs_2d (tools.py:2456-2749) re-stated on the
// height-map plus the placement history; used by macs.hip (stand-alone step)
// and transition.hip (fused step).
//
// The reference builds a list of "empty maximal spaces" (EMS), walks both bottom corners of every
// EMS sequentially -- sliding the block until it settles, with a `visited` set shared by all walks --
// scores every settled slot and takes the first maximum (optionally tie-broken by a usable-space
// score).  What is parallel here:
//   phase 1  EMS list: built by all lanes of the group redundantly in lock-step (identical LDS
//            writes, every lane reads back only what it wrote, so no barrier); the duplicate check of
//            block-top entries is strided over the lanes.
//   phase 2  the walks: whether a block settles at (x, Z) -- supported, free, and stable when the
//            reward is 'hard' -- depends on (x, Z) only, so a walk stops at the first position in its
//            direction that is `good` and not yet taken by an earlier walk.  Per EMS every lane tests
//            its own column once, ONE ballot gives the level's good-mask, and a walk is a
//            find-first-set on good & ~taken[Z] & range (`taken`: per-level bitmask in LDS).  This
//            phase is sequential over EMS but only integer bit-twiddling; it emits a slot list.
//   phase 3  scoring: lanes take slots round-robin and evaluate C/P/S in fp64 (and, only for slots
//            that tie at the maximum, the usable-space score); the reference's "first maximum, ties
//            by usable space, then by order" is one lexicographic butterfly reduction.
// Facts used: voxel (c,z) != 0 <=> z < hm[c]; level_free_space[z] == maximal runs of columns with
// hm[c] <= z (so only z = 0 and z in {hm[c]} open new level-EMS); the usable-space score of a
// candidate map hm' is base(hm') + (max_h - max(hm'))(W-1) with max_h common to all candidates, so
// ties are ordered by base(hm') - max(hm')(W-1).
#pragma once

#include "tap_place.h"

// EMS entries one step can produce: (W+1)/2 free runs on each of at most W+1 distinct levels, plus at
// most two per placed block (tools.py:2517-2555); two walks (slots) per EMS.  Even, for alignment.
__host__ __device__ constexpr int macs_ems_cap(int W, int n_max)
{
    const int cap = ((W + 1) * ((W + 1) / 2) + 2 * n_max + 1) & ~1;
    return cap < 16 ? 16 : cap;   // the slot list (2 cap ints) doubles as the 2 G-entry scratch of phase 1 (b), G <= 16
}
constexpr int MACS_MAX_H = 4096;   // the per-level `taken` masks live in LDS (2 bytes per level); EMS entries hold z in 15 bits

// LDS words per env group: hm | ems[cap] | slots[2 cap] | taken (uint16 per level) | history (x, z, bx, bz)
__host__ __device__ constexpr int macs_group_words(int G, int H, int n_max, int W)
{
    return G + 3 * macs_ems_cap(W, n_max) + (H + 1) / 2 + 4 * n_max;
}

struct MacsLds {
    int *hm, *ems, *slots, *hist;
    unsigned short *taken;
    int ems_cap;
};

__device__ __forceinline__ MacsLds macs_lds(int *base, int G, int H, int ems_cap)
{
    MacsLds m;
    m.hm = base;
    m.ems = base + G;
    m.ems_cap = ems_cap;
    m.slots = m.ems + ems_cap;
    m.taken = reinterpret_cast<unsigned short *>(m.slots + 2 * ems_cap);
    m.hist = m.slots + 2 * ems_cap + (H + 1) / 2;
    return m;
}


// usable-space tie-break score of the candidate map (hm with columns [xs, xs+bx) raised to `top`):
// sum_{h < m} longest free run (length - 1) at level h (tools.py:2667-2678), minus m (W - 1)
template <int G>
__device__ inline int macs_adj(const int (&hmr)[G], int W, int xs, int bx, int top, int m)
{
    int base = 0;
#pragma unroll
    for (int j = 0; j < G; ++j) {
        if (j >= W) break;
        const int v = (j >= xs && j < xs + bx) ? top : hmr[j];
        bool first = true;
        int next = m, best_run = 0, run = -1;
#pragma unroll
        for (int k = 0; k < G; ++k) {
            if (k >= W) break;
            const int hk = (k >= xs && k < xs + bx) ? top : hmr[k];
            if (hk == v && k < j) first = false;
            if (hk > v) next = min(next, hk);
            if (hk <= v) { ++run; best_run = max(best_run, run); } else run = -1;
        }
        if (first && v < m) base += (next - v) * best_run;
    }
    return base - m * (W - 1);
}

// column bitmasks of the height-map held in registers (static indexing only, no LDS, no cross-lane)
template <int G> __device__ __forceinline__ unsigned macs_mask_le(const int (&hmr)[G], unsigned wmask, int z)
{
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < G; ++k) m |= (unsigned)(hmr[k] <= z) << k;
    return m & wmask;
}
template <int G> __device__ __forceinline__ unsigned macs_mask_eq(const int (&hmr)[G], unsigned wmask, int z)
{
    unsigned m = 0;
#pragma unroll
    for (int k = 0; k < G; ++k) m |= (unsigned)(hmr[k] == z) << k;
    return m & wmask;
}
template <int G> __device__ __forceinline__ int macs_sum(const int (&hmr)[G], int xs, int bx)
{
    int s = 0;
#pragma unroll
    for (int k = 0; k < G; ++k) s += (k >= xs && k < xs + bx) ? hmr[k] : 0;
    return s;
}

// -DTAP_PROF: shader-clock deltas per phase, first lane of each workgroup (scripts/prof_macs2d.py reads them)
#ifdef TAP_PROF
static __device__ unsigned int tap_prof_m2[8192 * 8];
#define M2_PROF(i) do { const long long t_ = clock64(); if ((threadIdx.x) == 0 && blockIdx.x < 8192) tap_prof_m2[blockIdx.x * 8 + (i)] = (unsigned)(t_ - tp2_); tp2_ = t_; } while (0)
#define M2_PROF_BEGIN long long tp2_ = clock64()
#define M2_PROF_NOTE(i, v) tap_prof_m2[blockIdx.x * 8 + (i)] = (unsigned)(v)
#else
#define M2_PROF_NOTE(i, v) do { } while (0)
#define M2_PROF(i) do { } while (0)
#define M2_PROF_BEGIN do { } while (0)
#endif

// One placement.  Preconditions: L.hm[cell] = hm, L.taken[0..H) = 0, L.hist[0..4*cnt.count) =
// (x, z, bx, bz) of the earlier steps, all visible to the group (wave-level sync by the caller).
// do_step is group-uniform.  On return hm/cnt are updated and res describes the placement.
//
// Every lane keeps the whole height-map in registers and derives per-level column bitmasks
// (free: hm <= z, on: hm == z) from it, so runs, supports, footprint tests and stability are bit
// operations; LDS only holds the EMS list, the slot list and the per-level `taken` masks.
// WC != 0: the container's width known at compile time (instantiated for c4's W = 7: 15.85 -> 15.46 us per fused step).
template <int G, int WC = 0>
__device__ inline Placement tap_macs_place(const PlaceCfg &c, const MacsLds &L, int cell, int gl0,
                                           int &hm, Counters &cnt, int &err, int bx, int bz,
                                           bool do_step)
{
    const int W = WC ? WC : c.W, H = c.H, ems_cap = L.ems_cap;
    const bool incell = cell < W;
    Placement res = {0, 0, 0, 0, 0};
    if (!do_step) return res;
    M2_PROF_BEGIN;
    const int hard = c.flags & TAP_F_HARD;
    const int vol = bx * bz, step = cnt.count;
    const unsigned wmask = (1u << W) - 1u, gmask = (1u << G) - 1u;
    int hmr[G];
#pragma unroll
    for (int k = 0; k < G; ++k) hmr[k] = L.hm[k];
    int gmax = 0;
#pragma unroll
    for (int k = 0; k < G; ++k) gmax = max(gmax, hmr[k]);
    // Group-uniform column masks of a level come from the lanes' own columns: one compare per lane and a DPP OR
    // over the group (bits 0..15: hm <= z "free", bits 16..31: hm == z "on") -- 7 vector instructions instead of
    // the 6 G of a scan over the register copy, which phases 1 and 2 paid per level, per placed block and per EMS
    // (c4: 5 % of the fused step at n = 20).  The register copy stays for phase 3, where every lane scores its OWN slot.
    const int hx = incell ? hm : INT_MAX;                   // lanes beyond W: never free, never on
    const unsigned cbit = 1u << cell, cbit2 = 0x10001u << cell;
    auto level_masks = [&](int z) -> unsigned {
        return (unsigned)group_or<G>((int)(hx < z ? cbit : hx == z ? cbit2 : 0u));
    };

    // ---- phase 1: EMS list (identical on every lane of the group) ---------------------------------
    int n_ems = 0;
#define EMS_PUSH(x1, z, x2)                                                                  \
    do {                                                                                     \
        if (n_ems < ems_cap) L.ems[n_ems++] = ((x1) & 0xff) | (((x2) & 0xff) << 8) | ((z) << 16); \
        else err |= 16;                                                                      \
    } while (0)
    // (a) per-level free runs (tools.py:2517-2529); only z = 0 and z in {hm[c]} differ from below.  Level 0 is
    //     listed by the whole group (identical writes); every other level belongs to the lane of the FIRST column at
    //     that height, which lists the level's runs from its register copy of the map, and the lists are appended in
    //     level order (offset = entries of the owners of lower levels) -- the levels of the 8 containers of a wave
    //     used to be walked one after the other, each walk as long as the longest of the eight.
    if (bz <= H) {                                                            // :2519 at z = 0
        unsigned m = level_masks(0) & 0xffffu;
        while (m) {
            const int x1 = __ffs((int)m) - 1;
            const int len = __ffs((int)~(m >> x1)) - 1;                       // maximal run [x1, x1+len)
            const unsigned run = ((1u << len) - 1u) << x1;
            m &= ~run;
            if (x1 + bx > W) break;                                           // :2525
            EMS_PUSH(x1, 0, x1 + len - 1);                                    // :2529
        }
    }
    {
        const int RS = (W + 1) / 2;                                           // runs of one level, at most
        int *mine = L.slots + cell * RS, *keys = L.slots + G * RS;            // scratch inside the (unused) slot list
        const int z = hm;
        const unsigned on = macs_mask_eq<G>(hmr, wmask, z);
        const bool owner = incell && z > 0 && z + bz <= H && !(on & ((1u << cell) - 1u));   // :2519, first column at z
        int cn = 0;
        if (owner) {
            unsigned m = macs_mask_le<G>(hmr, wmask, z);
            while (m) {
                const int x1 = __ffs((int)m) - 1;
                const int len = __ffs((int)~(m >> x1)) - 1;
                const unsigned run = ((1u << len) - 1u) << x1;
                m &= ~run;
                if (x1 + bx > W) break;                                       // :2525
                if (!(on & run)) continue;                                    // :2526-2528 same run below
                mine[cn++] = (x1 & 0xff) | (((x1 + len - 1) & 0xff) << 8) | (z << 16);
            }
        }
        keys[cell] = owner ? ((z << 8) | cn) : -1;
        tap_wave_lds_sync();
        int off = 0, total = 0;
        for (int k = 0; k < G; ++k) {                                         // uniform addresses: broadcast reads
            const int kk = keys[k];
            if (kk < 0) continue;
            total += kk & 0xff;
            if ((kk >> 8) < z) off += kk & 0xff;
        }
        for (int j = 0; j < cn; ++j) {
            const int at = n_ems + off + j;
            if (at < ems_cap) L.ems[at] = mine[j]; else err |= 16;
        }
        n_ems = min(ems_cap, n_ems + total);
        tap_wave_lds_sync();
    }
    M2_PROF(0);
    // (b) tops of the blocks placed so far (tools.py:2531-2555); failed steps sit at (0, 0).
    //     What a placed block contributes depends on the block and the height-map only, so the blocks are taken ONE
    //     PER LANE, G at a time (round 3; the group-uniform loop cost 750 cycles per block, 44 % of a placement at
    //     step 19 of c4): the lane scans its register copy of the map at ITS block's top level, forms its (at most
    //     two) entries, and the entries are appended in block order through prefix counts.  A fully free top is
    //     appended only when the list does not hold it yet (:2538): absent from the list so far and from the entries
    //     of the earlier blocks of the same round (an equal earlier entry is in the list, or equals one that is).
    const unsigned below_me = (1u << cell) - 1u;
    for (int base = 0; base < step; base += G) {
        const int i = base + cell;
        const bool valid = i < step;
        const int hi4 = (valid ? i : base) * 4;
        const int x = L.hist[hi4], z = L.hist[hi4 + 1], xx = L.hist[hi4 + 2], zz = L.hist[hi4 + 3];
        const int tz = z + zz;
        int c0 = -1, c1 = -1;
        bool flagged = false;
        if (valid && tz < H) {                                                // :2535
            const unsigned fr = macs_mask_le<G>(hmr, wmask, tz);
            const unsigned span = (xx >= 32 ? 0xffffffffu : ((1u << xx) - 1u)) << x; // slice clips at W (:2537)
            const int xe = x + xx - 1;
            if (((span & wmask) & ~fr) == 0) {
                c0 = (x & 0xff) | ((xe & 0xff) << 8) | (tz << 16);
                flagged = true;
            } else if (xe >= W) {
                err |= 8;                                                     // reference: IndexError :2550
            } else {
                if (((fr >> x) & 1u) && x > 0 && ((fr >> (x - 1)) & 1u)) {    // :2543-2548 left part
                    const int len = __ffs((int)~(fr >> x)) - 1;               // free columns from x rightwards
                    c0 = (x & 0xff) | (((x + min(len, xx) - 1) & 0xff) << 8) | (tz << 16);
                }
                if (((fr >> xe) & 1u) && x + xx < W && ((fr >> (x + xx)) & 1u)) { // :2550-2555 right part
                    const unsigned low = fr << (31 - xe);                     // bit xe -> bit 31
                    const int len = __clz((int)~low);                         // free columns from xe leftwards
                    c1 = ((xe - min(len, xx) + 1) & 0xff) | ((xe & 0xff) << 8) | (tz << 16);
                }
            }
        }
        // this round's entries side by side in the (still unused) slot list, for the "already listed" test
        L.slots[2 * cell] = c0;
        L.slots[2 * cell + 1] = c1;
        tap_wave_lds_sync();
        bool keep0 = c0 >= 0;
        if (flagged) {
            bool dup = false;
            for (int k = 0; k < n_ems; ++k) dup |= L.ems[k] == c0;
            for (int k = 0; k < 2 * cell; ++k) dup |= L.slots[k] == c0;
            keep0 = !dup;
        }
        const bool keep1 = c1 >= 0;
        const unsigned k0 = (unsigned)((__ballot(keep0) >> gl0) & gmask), k1 = (unsigned)((__ballot(keep1) >> gl0) & gmask);
        const int at0 = n_ems + __popc(k0 & below_me) + __popc(k1 & below_me), at1 = at0 + (keep0 ? 1 : 0);
        if (keep0) { if (at0 < ems_cap) L.ems[at0] = c0; else err |= 16; }
        if (keep1) { if (at1 < ems_cap) L.ems[at1] = c1; else err |= 16; }
        n_ems = min(ems_cap, n_ems + __popc(k0) + __popc(k1));
        tap_wave_lds_sync();                                                  // entries written by other lanes
    }

    M2_PROF(1);
    // ---- phase 2: both corner walks of every EMS (tools.py:2680-2700) -> slot list ------------------
    const int X = W - bx + 1;
    const unsigned fpm = (1u << bx) - 1u;
    int n_slots = 0;
    int pk_next = n_ems > 0 ? L.ems[0] : 0;
    for (int e = 0; e < n_ems; ++e) {
        const int pk = pk_next;
        pk_next = L.ems[e + 1 < n_ems ? e + 1 : e];                           // the next entry, in flight under this one
        const int X1 = pk & 0xff, X2 = (pk >> 8) & 0xff, Z = pk >> 16;
        // every lane tests its own column as the block's left edge at level Z (:2571-2588)
        const unsigned lm = level_masks(Z), fr = lm & 0xffffu, on = lm >> 16;
        bool good = false;
        if (incell && cell + bx <= W) {
            const unsigned eq = (on >> cell) & fpm;
            const bool free_ = ((fr >> cell) & fpm) == fpm;                   // :2576
            const bool supported = Z == 0 || eq != 0;                         // :2574
            const int stab = (Z == 0) ? 1 : (eq ? tap_stable2d(bx, eq) : 0);  // :2577-2585
            good = supported && free_ && (stab || !hard);                     // :2580-2581
        }
        const unsigned gm = (unsigned)((__ballot(good) >> gl0) & gmask);
        unsigned tk = L.taken[Z];
        if (X1 < X) {                                                         // :2686 left corner, slide right
            const unsigned m = gm & ~tk & ~((1u << X1) - 1u);
            if (m) {
                const int xs = __ffs((int)m) - 1;
                tk |= 1u << xs;
                L.slots[n_slots++] = xs | (Z << 8);
            }
        }
        const int hi = X2 - bx + 1;                                           // :2694 right corner, slide left
        if (hi >= 0) {
            if (hi + bx > W) err |= 8;
            else {
                const unsigned m = gm & ~tk & ((2u << hi) - 1u);
                if (m) {
                    const int xs = 31 - __clz((int)m);
                    tk |= 1u << xs;
                    L.slots[n_slots++] = xs | (Z << 8);
                }
            }
        }
        L.taken[Z] = (unsigned short)tk; // every lane stores the same value and reads back its own
    }

    M2_PROF(2);
    // ---- phase 3: score the slots (tools.py:2590-2604), lanes round-robin ---------------------------
    const int valid2 = cnt.valid + vol;
    const bool tiebreak = (c.flags & TAP_F_MCS_TIE) != 0, zero = (c.flags & TAP_F_MCS_ZERO) != 0;
    // C/P/S of slot s -> ratio (0.0 when the reward string zeroes it, :2709-2710)
    auto eval_slot = [&](int s, int &xs, int &Z, int &sum, int &stab) -> double {
        const int sp = L.slots[s];
        xs = sp & 0xff; Z = sp >> 8;
        sum = macs_sum<G>(hmr, xs, bx);
        const unsigned eq = (macs_mask_eq<G>(hmr, wmask, Z) >> xs) & fpm;     // a settled slot has max == Z
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
    M2_PROF(3);
    // winner (:2713-2736): the first slot reaching rmax, or -- with the 'mcs' tie-break -- the first
    // one among them with the largest usable-space score
    int win = INT_MAX;
    if (n_slots > 0) {
        if (!tiebreak) {
            win = group_min<G>(my_r == rmax ? my_slot : INT_MAX);
        } else {
            int best_adj = INT_MIN, best_s = INT_MAX, n_tied = 0, max_height = gmax;
            for (int s = cell; s < n_slots; s += G) {
                int xs, Z, sum, stab;
                const double r = eval_slot(s, xs, Z, sum, stab);              // recomputing beats storing r
                max_height = max(max_height, Z + bz);                         // :2719 np.max(heightmap_ems)
                if (r != rmax) continue;
                ++n_tied;
                const int adj = macs_adj<G>(hmr, W, xs, bx, Z + bz, max(gmax, Z + bz));
                if (adj > best_adj) { best_adj = adj; best_s = s; }
            }
            group_butterfly<G>((int)(threadIdx.x & 63), [&](auto get) { // lexicographic (adj desc, order asc)
                const int a2 = get(best_adj), s2 = get(best_s);
                if (a2 > best_adj || (a2 == best_adj && s2 < best_s)) { best_adj = a2; best_s = s2; }
            });
            win = best_s;
            // the reference only scores usable space when more than one entry ties (:2718), and then
            // indexes levels up to max_height: IndexError once any settled slot reaches above H
            const int nt = zero ? 2 * n_ems : group_sum<G>(n_tied);
            if (nt > 1 && group_max<G>(max_height) > H) err |= 1;
        }
    }

    M2_PROF(4);
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
    M2_PROF(5);
    if ((threadIdx.x) == 0 && blockIdx.x < 8192) { M2_PROF_NOTE(6, n_ems); M2_PROF_NOTE(7, n_slots); }
#undef EMS_PUSH
    return res;
}
