// tap_waves.h -- the "placement wave" shared by the fused kernels (transition.hip, rolling.hip):
// the lanes of one wavefront carry 64/G lane-per-cell groups, each stepping one container with
// tools.Container.add_new_block (LB_GREEDY) on a block given directly or gathered from `static`
// (model.py:404-465),
// optionally from a fresh container and optionally emitting calc_ratio (model.py:499-510).
#pragma once

#include "tap_common.h"
#include "tap_place.h"
#include "tap_masks.h"

// env: this group's container; cell / lane: lane index inside the group / the wave;
// g_old, g_new: the group's two G-int LDS slices.  Every lane of the wave must call this.
template <int D, int G, bool HARDOK = true>
__device__ __forceinline__ void tap_lb_place_wave(const StepArgs &s, int flags, float *ratio_out,
                                                  int env, int cell, int lane, int *g_old, int *g_new)
{
    const int W = s.d.W, L = s.d.L, cells = W * L;
    const bool fresh = flags & TAP_T_FRESH;
    const int B = s.d.B;
    const bool ev = env < B, incell = cell < cells;
    // Every load is unconditional on a clamped address (an exec-masked `cond ? p[i] : 0` keeps the compiler from
    // moving loads across it) and `ptr` -- the head of the step's only dependent chain, ptr -> static[ptr] -- goes
    // out first; in the fused kernels its address comes from preloaded kernel arguments (transition.hip).
    const int envc = ev ? env : 0;
    long praw = 0;
    if (s.static_) praw = (long)s.ptr[envc];
    const int hm_l = s.v.hm[(size_t)envc * cells + min(cell, cells - 1)];
    const int cv_l = s.v.cnt[(size_t)envc * 4 + (cell & 3)];
    unsigned char act_l = 1;
    if (s.active) act_l = s.active[envc];
    const int hm0 = (ev && !fresh && incell) ? hm_l : 0;
    const int cv = (ev && !fresh && cell < 4) ? cv_l : 0;
    int hm = hm0, dims[3] = {1, 1, 1};
    float fv[3] = {0.f, 0.f, 0.f};   // the gathered sides as floats: decoder_static, written with the step's other results
    if (s.static_) { // gather of model.py:404-412
        bool badp;
        const long p = tap_col(praw, s.nR, badp);
        for (int k = 0; k < D; ++k) { // unconditional load of a valid column, then the select
            const float v = s.static_[((size_t)envc * s.static_rows + 1 + k) * s.nR + p];
            dims[k] = !ev ? 1 : badp ? 0 : (int)v;
            fv[k] = badp ? 0.f : v;
        }
    } else if (s.blocks_dtype == TAP_DT_F32) { // block.astype(int), tools.py:3689
        for (int k = 0; k < D; ++k) { const float v = ((const float *)s.blocks)[(size_t)envc * D + k]; dims[k] = ev ? (int)v : 1; }
    } else {
        for (int k = 0; k < D; ++k) { const int v = ((const int32_t *)s.blocks)[(size_t)envc * D + k]; dims[k] = ev ? v : 1; }
    }
    // `active` = 0: the 'mul' input types' idle container -- untouched, only reports its feature
    const bool act = ev && act_l != 0;
    const int gl0 = lane - cell;
    Counters cnt = {__shfl(cv, gl0), __shfl(cv, gl0 + 1), __shfl(cv, gl0 + 2), __shfl(cv, gl0 + 3)};
    const int bx = dims[0], by = D == 3 ? dims[1] : 1, bz = dims[D - 1];
    int err = 0;
    bool do_step = act;
    if (act && cnt.count >= s.d.n_max) { err |= 2; do_step = false; }  // tools.py:3677 IndexError
    if (act && (bx < 1 || by < 1 || bz < 1)) { err |= 4; do_step = false; }
    g_old[cell] = hm;
#ifdef TAP_PROF
    TL_WAIT_VM(); TL_STAMP(1);
#endif
    tap_wave_lds_sync();
    const PlaceCfg cfg = {W, L, s.d.H, s.d.flags, s.lut};
    const int step = cnt.count;
    const Placement pl = tap_place<D, G, HARDOK>(cfg, g_old, cell, hm, cnt, err, bx, by, bz, do_step);
    err = group_or<G>(err);
    g_new[cell] = hm;
    TL_STAMP(2);
    tap_wave_lds_sync();
    const int gmax = (flags & TAP_T_RATIO) ? group_max<G>(incell ? hm : 0) : 0;
    if (ev) {
        if (incell) s.v.hm[(size_t)env * cells + cell] = hm;
        if (s.feature_out)
            tap_write_feature<D, G>(s.d.feature, W, L, g_new, cell, hm,
                                    s.feature_out + (size_t)env * s.flen);
        if (cell == 0) {
            if (s.static_) tap_step_aux(s, env, D, fv, praw);   // after the placement: no load of this wave waits behind these stores
            if (do_step || fresh)
                reinterpret_cast<int4 *>(s.v.cnt)[env] = make_int4(cnt.valid, cnt.empty, cnt.nstable, cnt.count);
            if (do_step) {
                int32_t *q = s.v.pos + (size_t)step * D * B + env;
                q[0] = pl.x;
                if (D == 3) { q[B] = pl.y; q[2 * (size_t)B] = pl.z; } else q[B] = pl.z;
                s.v.stable[(size_t)step * B + env] = (uint8_t)pl.stab;
            }
            if (fresh) s.v.err[env] = err;
            else if (err) s.v.err[env] |= err;
            if (flags & TAP_T_RATIO) { // tools.py:3887-3966 on the state just written
                double C = 0.0, P = 0.0, S = 0.0;
                if (cnt.count != 0) {
                    C = (double)cnt.valid / (double)((long long)W * L * gmax);
                    P = (double)cnt.valid / (double)(cnt.empty + cnt.valid);
                    S = (double)cnt.nstable / (double)cnt.count;
                }
                ratio_out[env] = (float)tap_ratio_formula(s.d.ratio_mode, C, P, S);
            }
        }
    } else if (s.d.feature == TAP_FEAT_ZERO) {
        (void)group_min<G>(INT_MAX);
    }
}

