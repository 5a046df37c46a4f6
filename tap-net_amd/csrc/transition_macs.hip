// transition_macs.hip -- the fused step of transition.hip (update_dynamic + update_mask + gather + add_new_block in one
// launch, pack.py:276-376, model.py:404-465) for the MACS / MUL placements: 2D up to 16 columns (tap_macs.h) and 3D
// (tap_macs3.h).  Same workgroup shape as k_transition: the envs' placement waves at raised priority beside 4 stream
// waves.  Its own translation unit so that it compiles beside transition.hip (the two were the build's long pole).
// gfx950 only.
#include <cstdlib>

#include "tap_common.h"
#include "tap_macs.h"
#include "tap_macs3.h"
#include "tap_masks.h"
#include "tap_place.h"
#include "tap_transition.h"
#include "tap_waves.h"

// ---- the same fusion for MACS / MUL 2D (tap_macs.h): G = 8/16 lanes per env ---------------------
// (Register cliff, measured in round 3: a workgroup is 5 waves and a CU gets 4 of them at B = 8192; at 71 VGPRs -- 7
//  waves per SIMD -- the step takes 17.8 us, at 85 -- 5 per SIMD, where the 20 waves only fit if the dispatcher spreads
//  them perfectly -- 23.6 us.  A window-maximum form of macs_adj (80 instead of 590 instructions, tie-break 5.1 k -> 2.6 k
//  cycles stand-alone) crossed that line and was dropped; amdgpu_waves_per_eu(7, 8) brings either form to 72 VGPRs but
//  costs 3 % by itself.)
#ifndef TAP_MACS_SW
#define TAP_MACS_SW 4   // stream waves per workgroup of the MACS fused steps (-DTAP_MACS_SW=8, measured in round 5: c4 503 -> 330 M
                        // env-steps/s, c6 140 -> 129 M -- the wave slots are worth more to the placement waves)
#endif
// WC: 0, or the container's width known at compile time (tap_macs_place; BASELINE configs[3]'s W = 7 with its n = 20 window)
template <int G, int NC, int MODE, int WC = 0>
__global__ void __launch_bounds__((TransGeom<G, TAP_MACS_SW>::THREADS)) k_transition_macs(TransArgs a)
{
    using Geo = TransGeom<G, TAP_MACS_SW>;
    constexpr int EPB = Geo::EPB, SPW = Geo::SPW, ENV_WAVES = Geo::ENV_WAVES;
    extern __shared__ float trans_lds[];
    // (the wave index stays a vector value here: in a scalar register -- transition.hip, where it is worth 4 % at c2 -- it
    //  measured flat on the MACS steps and 7 % SLOWER on the rolling step, round 6)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int env_base = blockIdx.x * EPB;
    const int B = a.s.d.B, W = WC ? WC : a.s.d.W, H = a.s.d.H;
    if (wave >= ENV_WAVES) {
        trans_stream_wave<SPW, NC, MODE>(a.m, env_base + (wave - ENV_WAVES) * SPW, lane,
                                     trans_lds + (size_t)(wave - ENV_WAVES) * SPW * 3 * a.m.nR);
        return;
    }
    __builtin_amdgcn_s_setprio(3);   // (2 -> 3: +2 % at c4; without a raised priority the step takes 23.5 instead of 18.0 us)
    int *macs_base = reinterpret_cast<int *>(trans_lds + (size_t)EPB * 3 * a.m.nR);
    const bool fresh = a.flags & TAP_T_FRESH;
    const int cell = tid % G, gl0 = lane - cell;
    const int env = env_base + tid / G;
    const bool ev = env < B, incell = cell < W;
    const MacsLds L = macs_lds(macs_base + (tid / G) * macs_group_words(G, H, a.s.d.n_max, W), G, H, macs_ems_cap(W, a.s.d.n_max));
    int hm = 0, cv = 0, bx = 1, bz = 1;
    if (ev) {
        if (!fresh) {
            if (incell) hm = a.s.v.hm[(size_t)env * W + cell];
            if (cell < 4) cv = a.s.v.cnt[(size_t)env * 4 + cell];
        }
        bool badp;
        const long praw = (long)a.s.ptr[env];
        const long p = tap_col(praw, a.s.nR, badp);
        const float vx = a.s.static_[((size_t)env * a.s.static_rows + 1) * a.s.nR + p];
        const float vz = a.s.static_[((size_t)env * a.s.static_rows + 2) * a.s.nR + p];
        bx = badp ? 0 : (int)vx;
        bz = badp ? 0 : (int)vz;
        if (cell == 0) { const float fv[2] = {badp ? 0.f : vx, badp ? 0.f : vz}; tap_step_aux(a.s, env, 2, fv, praw); }
    }
    Counters cnt = {__shfl(cv, gl0), __shfl(cv, gl0 + 1), __shfl(cv, gl0 + 2), __shfl(cv, gl0 + 3)};
    int err = 0;
    bool do_step = ev;
    if (ev && cnt.count >= a.s.d.n_max) { err |= 2; do_step = false; }
    if (ev && (bx < 1 || bz < 1)) { err |= 4; do_step = false; }
    L.hm[cell] = hm;
    for (int i = cell; i < H; i += G) L.taken[i] = 0;
    if (ev)
        for (int k = cell; k < cnt.count * 4 && k < a.s.d.n_max * 4; k += G) {
            const int i = k >> 2, f = k & 3;
            L.hist[k] = (f < 2 ? a.s.v.pos : a.s.v.blk)[(size_t)(i * 2 + (f & 1)) * B + env];
        }
    tap_wave_lds_sync();
    const int step = cnt.count;
    const PlaceCfg cfg = {W, 1, H, a.s.d.flags, nullptr};
    const Placement pl = tap_macs_place<G, WC>(cfg, L, cell, gl0, hm, cnt, err, bx, bz, do_step);
    err = group_or<G>(err);
    tap_wave_lds_sync();
    L.hm[cell] = hm;
    tap_wave_lds_sync();
    const int gmax = (a.flags & TAP_T_RATIO) ? group_max<G>(incell ? hm : 0) : 0;
    if (ev) {
        if (incell) a.s.v.hm[(size_t)env * W + cell] = hm;
        if (a.s.feature_out)
            tap_write_feature<2, G>(a.s.d.feature, W, 1, L.hm, cell, hm, a.s.feature_out + (size_t)env * a.s.flen);
        if (cell == 0) {
            if (do_step || fresh)
                reinterpret_cast<int4 *>(a.s.v.cnt)[env] = make_int4(cnt.valid, cnt.empty, cnt.nstable, cnt.count);
            if (do_step) {
                a.s.v.pos[(size_t)(step * 2) * B + env] = pl.x;
                a.s.v.pos[(size_t)(step * 2 + 1) * B + env] = pl.z;
                a.s.v.stable[(size_t)step * B + env] = (uint8_t)pl.stab;
                a.s.v.blk[(size_t)(step * 2) * B + env] = bx;
                a.s.v.blk[(size_t)(step * 2 + 1) * B + env] = bz;
            }
            if (fresh) a.s.v.err[env] = err;
            else if (err) a.s.v.err[env] |= err;
            if (a.flags & TAP_T_RATIO) {
                double C = 0.0, P = 0.0, S = 0.0;
                if (cnt.count != 0) {
                    C = (double)cnt.valid / (double)((long long)W * gmax);
                    P = (double)cnt.valid / (double)(cnt.empty + cnt.valid);
                    S = (double)cnt.nstable / (double)cnt.count;
                }
                a.ratio_out[env] = (float)tap_ratio_formula(a.s.d.ratio_mode, C, P, S);
            }
        }
    } else if (a.s.d.feature == TAP_FEAT_ZERO) {
        (void)group_min<G>(INT_MAX);
    }
}

// ---- and for MACS / MUL 3D (tap_macs3.h): G = 8..64 lanes per env --------------------------------
// -DTAP_M3_SPREAD (A/B builds): ONE container per placement wave at G = 32 -- the wave's second lane group idles on an
// out-of-range env -- so that B = 4096 puts four placement waves on a SIMD instead of two, each running only its own
// container's loop trips.
// Loop form of the fp32 expansion in the MACS steps' stream waves: 5 / 6 = shadow (given / built) | TAP_MODE_MERGED, the
// run-of-rows loop; 1 / 2 = slab by slab.  Round 6, same session, two runs each (profiles/r06_macs_merge_ab.txt), M env-steps/s,
// run-of-rows against slab-by-slab: MACS 2D (c4's shape) 518.6 / 518.0 against 508.7 / 511.8 at B = 8 192, 461.8 / 482.5
// against 467.7 / 482.8 at 32 768 (nontemporal stores from here on), 533.4 / 526.0 against 518.8 / 519.6 at 131 072 -- the
// run-of-rows loop at every batch, unlike the LB_GREEDY step (transition.hip), whose 2D windows lose 15-20 % with it once
// the stores are nontemporal; MACS 3D (c6's shape, nR = 60) 139.8 / 139.7 against 139.9 / 140.1 at B = 4 096, 195.2 / 195.6
// against 197.2 / 197.8 at 32 768, 222.9 / 222.9 against 225.8 / 225.8 at 131 072 -- slab by slab, as for the LB_GREEDY
// step's 3D windows.  -DTAP_MACS_NOMERGE / -DTAP_MACS_MERGE_ALL force one form everywhere (A/B builds).
#if defined(TAP_MACS_NOMERGE)
#define TAP_MACS_M1 1
#define TAP_MACS_M2 2
#define TAP_MACS3_M1 1
#define TAP_MACS3_M2 2
#elif defined(TAP_MACS_MERGE_ALL)
#define TAP_MACS_M1 5
#define TAP_MACS_M2 6
#define TAP_MACS3_M1 5
#define TAP_MACS3_M2 6
#else
#define TAP_MACS_M1 5
#define TAP_MACS_M2 6
#define TAP_MACS3_M1 1
#define TAP_MACS3_M2 2
#endif
#ifdef TAP_M3_SPREAD
template <int G> struct M3Spread { static constexpr bool on = G == 32; };
#else
template <int G> struct M3Spread { static constexpr bool on = false; };
#endif
template <int G> struct M3Geo {
    using Geo = TransGeom<G, TAP_MACS_SW>;
    static constexpr bool SPREAD = M3Spread<G>::on;
    static constexpr int EPB = SPREAD ? 4 : Geo::EPB, SPW = SPREAD ? 2 : Geo::SPW, ENV_WAVES = SPREAD ? 4 : Geo::ENV_WAVES;
    static constexpr int GROUPS = SPREAD ? 8 : Geo::EPB;                      // LDS regions (the idle halves get one too)
    static constexpr int THREADS = SPREAD ? 384 : Geo::THREADS;              // 4 placement + 2 stream waves
};

#ifdef TAP_M3_CAP6
#define M3_OCC __attribute__((amdgpu_waves_per_eu(6, 6)))
#else
#define M3_OCC
#endif
// WL: 0, or the side of a WL x WL container known at compile time (tap_macs3_place; instantiated for the reference's 5 x 5)
template <int G, int NC, int MODE, int WL = 0>
__global__ void __launch_bounds__((M3Geo<G>::THREADS)) M3_OCC k_transition_macs3(TransArgs a)
{
    using Geo = M3Geo<G>;
    constexpr int EPB = Geo::EPB, SPW = Geo::SPW, ENV_WAVES = Geo::ENV_WAVES;
    extern __shared__ float trans_lds[];
    // (the wave index stays a vector value here: in a scalar register -- transition.hip, where it is worth 4 % at c2 -- it
    //  measured flat on the MACS steps and 7 % SLOWER on the rolling step, round 6)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int env_base = blockIdx.x * EPB;
    if (wave >= ENV_WAVES) {
        trans_stream_wave<SPW, NC, MODE>(a.m, env_base + (wave - ENV_WAVES) * SPW, lane,
                                     trans_lds + (size_t)(wave - ENV_WAVES) * SPW * 3 * a.m.nR);
        return;
    }
    __builtin_amdgcn_s_setprio(2);
    int *macs_base = reinterpret_cast<int *>(trans_lds + (size_t)EPB * 3 * a.m.nR);
    const int grp = tid / G;
    const int env = Geo::SPREAD ? ((grp & 1) ? a.s.d.B : env_base + (grp >> 1)) : env_base + grp;
    tap_macs3_wave<G, WL>(a.s, a.flags, a.ratio_out, env, tid % G, lane,
                          macs_base + grp * macs3_group_words(G, a.s.d.n_max, a.s.d.H));
}

template <int G> static int launch_transition_macs3(tap_ctx *ctx, const TransArgs &a, hipStream_t st)
{
    constexpr int EPB = M3Geo<G>::EPB, THREADS = M3Geo<G>::THREADS;
    const int grid = (a.s.d.B + EPB - 1) / EPB;
    if (grid == 0) return TAP_OK;
    const size_t lds = (size_t)EPB * 3 * a.m.nR * sizeof(float) +
                       (size_t)M3Geo<G>::GROUPS * macs3_group_words(G, a.s.d.n_max, a.s.d.H) * sizeof(int);
    if (lds > tap_lds_limit(ctx)) return tap_fail(ctx, TAP_E_UNSUPPORTED, "transition(MACS 3D): %zu bytes of LDS needed", lds);
    const int mode = a.m.bits_in ? 1 : mask_builds_bits(a.m) ? 2 : 0;
    // the reference's own 3D container (5 x 5, BASELINE c6; a 32-lane group) runs the instantiation with compile-time sides
    const bool wl5 = G == 32 && a.s.d.W == 5 && a.s.d.L == 5;
#define TAP_LAUNCH_W(NC_, M_, LDS_, WL_) do { TAP_HIP_CHECK(ctx, tap_allow_lds(k_transition_macs3<G, NC_, M_, WL_>, LDS_)); \
        hipLaunchKernelGGL((k_transition_macs3<G, NC_, M_, WL_>), dim3(grid), dim3(THREADS), LDS_, st, a); } while (0)
#define TAP_LAUNCH_T(NC_, M_, LDS_) do { if constexpr (G == 32) { if (wl5) TAP_LAUNCH_W(NC_, M_, LDS_, 5); else TAP_LAUNCH_W(NC_, M_, LDS_, 0); } \
        else TAP_LAUNCH_W(NC_, M_, LDS_, 0); } while (0)
    const bool inpl = mode == 1 && a.m.inplace && a.m.dyn_out;        // tap_masks.h: MaskArgs::inplace
#define TAP_LAUNCH_M(NC_, LDS_) do { if (inpl) TAP_LAUNCH_T(NC_, (1 | TAP_MODE_INPLACE), LDS_); else if (mode == 1) TAP_LAUNCH_T(NC_, TAP_MACS3_M1, LDS_); else if (mode == 2) TAP_LAUNCH_T(NC_, TAP_MACS3_M2, LDS_); else TAP_LAUNCH_T(NC_, 0, LDS_); } while (0)
    switch (mask_fast_path_cols(a.m)) {
    case 1: TAP_LAUNCH_M(1, lds); break;
    case 2: TAP_LAUNCH_M(2, lds); break;
    case 4: TAP_LAUNCH_M(4, lds); break;
    default: TAP_LAUNCH_T(0, 0, lds); break;
    }
#undef TAP_LAUNCH_M
#undef TAP_LAUNCH_T
#undef TAP_LAUNCH_W
    (void)wl5; (void)inpl;
    TAP_LAUNCH_CHECK(ctx, "k_transition_macs3");
    return TAP_OK;
}

int tap_macs_validate(tap_ctx *ctx, const tap_env_desc &d); // macs.hip

template <int G> static int launch_transition_macs(tap_ctx *ctx, const TransArgs &a, hipStream_t st)
{
    constexpr int EPB = TransGeom<G, TAP_MACS_SW>::EPB, THREADS = TransGeom<G, TAP_MACS_SW>::THREADS;
    const int grid = (a.s.d.B + EPB - 1) / EPB;
    if (grid == 0) return TAP_OK;
    const size_t lds = (size_t)EPB * 3 * a.m.nR * sizeof(float) +
                       (size_t)EPB * macs_group_words(G, a.s.d.H, a.s.d.n_max, a.s.d.W) * sizeof(int);
    if (lds > tap_lds_limit(ctx)) return tap_fail(ctx, TAP_E_UNSUPPORTED, "transition(MACS): %zu bytes of LDS needed", lds);
    const int mode = a.m.bits_in ? 1 : mask_builds_bits(a.m) ? 2 : 0;
    // BASELINE configs[3] (c4: W = 7, windows of 20 nodes) on the bit shadow runs the instantiation with the width and
    // the window's shape compiled in
    const bool c4shape = G == 8 && a.s.d.W == 7 && tap_mode_shape20_ok(a.m);
#define TAP_LAUNCH_K(NC_, M_, LDS_, WC_) do { TAP_HIP_CHECK(ctx, tap_allow_lds(k_transition_macs<G, NC_, M_, WC_>, LDS_)); \
        hipLaunchKernelGGL((k_transition_macs<G, NC_, M_, WC_>), dim3(grid), dim3(THREADS), LDS_, st, a); } while (0)
#define TAP_LAUNCH_T(NC_, M_, LDS_) do { if constexpr (G == 8 && (NC_) == 1 && ((M_) & 3) != 0) { \
            if (c4shape) TAP_LAUNCH_K(NC_, ((M_) | TAP_MODE_C4_10), LDS_, 7); else TAP_LAUNCH_K(NC_, M_, LDS_, 0); } \
        else TAP_LAUNCH_K(NC_, M_, LDS_, 0); } while (0)
    const bool inpl = mode == 1 && a.m.inplace && a.m.dyn_out;        // tap_masks.h: MaskArgs::inplace
#define TAP_LAUNCH_M(NC_, LDS_) do { if (inpl) TAP_LAUNCH_T(NC_, (1 | TAP_MODE_INPLACE), LDS_); else if (mode == 1) TAP_LAUNCH_T(NC_, TAP_MACS_M1, LDS_); else if (mode == 2) TAP_LAUNCH_T(NC_, TAP_MACS_M2, LDS_); else TAP_LAUNCH_T(NC_, 0, LDS_); } while (0)
    switch (mask_fast_path_cols(a.m)) {
    case 1: TAP_LAUNCH_M(1, lds); break;
    case 2: TAP_LAUNCH_M(2, lds); break;
    case 4: TAP_LAUNCH_M(4, lds); break;
    default: TAP_LAUNCH_T(0, 0, lds); break;
    }
#undef TAP_LAUNCH_M
#undef TAP_LAUNCH_T
#undef TAP_LAUNCH_K
    (void)c4shape; (void)inpl;
    TAP_LAUNCH_CHECK(ctx, "k_transition_macs");
    return TAP_OK;
}


int tap_transition_macs_launch(tap_ctx *ctx, const tap_env_desc *d, const TransArgs &a0, hipStream_t st)
{
    // these launches last 17 us and more, so the fp32 expansion's stores stay write-through up to a higher limit than
    // the LB_GREEDY step's (tap_masks.h: store_stream; c4's 78.6 MB per launch: 480 against 441 M env-steps/s)
    TransArgs a = a0;
    a.m.wt = tap_write_through((size_t)a.m.B * a.m.rows * a.m.nR * sizeof(float), (size_t)128 << 20);
    const int Gs = tap_group_size(d);
    if (d->D == 3) {
        switch (Gs) {
        case 8: return launch_transition_macs3<8>(ctx, a, st);
        case 16: return launch_transition_macs3<16>(ctx, a, st);
        case 32: return launch_transition_macs3<32>(ctx, a, st);
        default: return launch_transition_macs3<64>(ctx, a, st);
        }
    }
    // (16 lanes per container for W <= 8 measured 2.3 x slower at c4: the per-lane work grows with G)
    return d->W <= 8 ? launch_transition_macs<8>(ctx, a, st) : launch_transition_macs<16>(ctx, a, st);
}
