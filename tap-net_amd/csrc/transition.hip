// transition.hip -- one whole lock-step of DRL.forward's environment work in a single launch:
// update_dynamic + update_mask (pack.py:276-376, model.py:376-386) and the gather + add_new_block
// of model.py:404-465, optionally starting from a fresh container (Container.__init__) and
// optionally emitting calc_ratio (model.py:499-510).  gfx950 only.
//
// Why fuse: at the BASELINE batch sizes the placement moves ~0.9 MB per launch and is pure latency
// (two dependent loads, fp64 divides), while the precedence update streams ~40 MB and is HBM-bound.
// In one launch the placement's latency hides under the stream and a kernel boundary per step
// disappears.  A workgroup owns EPB = 8 consecutive envs (4 when an env needs 64 lanes) and is made
// of two kinds of waves: 4 stream waves, each copying EPB/4 dynamic slabs with 16-byte accesses
// (the slabs' loads interleaved so all are in flight together) and then updating their column sums
// and masks, and EPB*G/64 placement waves carrying the
// lane-per-cell groups (at raised priority, so their short dependent chain issues promptly).  The
// two kinds never exchange data, so there is no s_barrier: lane groups never span a wave and LDS
// hand-offs only need compiler ordering.  (A first version that let the same waves stream and then
// place showed no gain: every workgroup was in the same phase at the same time.)
#include <cstdlib>

#include "tap_common.h"
#include "tap_masks.h"
#include "tap_place.h"

struct TransArgs {
    StepArgs s;   // placement (always the gather form: s.static_, s.ptr)
    MaskArgs m;   // precedence update
    int flags;
    float *ratio_out;
};

template <int G, int SW> struct TransGeom {
    static constexpr int EPB = (G == 64) ? 4 : 8;   // envs per workgroup
    static constexpr int ENV_WAVES = EPB * G / 64;  // waves made of placement lane groups
    static constexpr int STREAM_WAVES = (SW < EPB) ? SW : EPB; // waves that stream the dynamic slabs
    static constexpr int SPW = EPB / STREAM_WAVES;  // slabs per stream wave
    static constexpr int THREADS = 64 * (ENV_WAVES + STREAM_WAVES);
};

template <int D, int G, bool FAST, int SW>
__global__ void __launch_bounds__((TransGeom<G, SW>::THREADS)) k_transition(TransArgs a)
{
    using Geo = TransGeom<G, SW>;
    constexpr int EPB = Geo::EPB, SPW = Geo::SPW, ENV_WAVES = Geo::ENV_WAVES;
    extern __shared__ float trans_lds[];
    __shared__ int s_old[64 * ENV_WAVES];
    __shared__ int s_new[64 * ENV_WAVES];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int env_base = blockIdx.x * EPB;
    const int B = a.s.d.B;

    if (wave >= ENV_WAVES) {
        // ---- stream waves: out-of-place copy of their slabs with the chosen rows cleared
        //      (pack.py:370-374), then the column sums + both masks (pack.py:318-329)
        const int sw = wave - ENV_WAVES;
        const int senv0 = env_base + sw * SPW;
        bool on[SPW];
#pragma unroll
        for (int k = 0; k < SPW; ++k) on[k] = senv0 + k < B;
        if (FAST) {
            stream_wave_fast<SPW, 6>(a.m, senv0, lane, on, trans_lds + (size_t)sw * SPW * 3 * a.m.nR);
        } else {
            const size_t slab = (size_t)a.m.rows * a.m.nR;
#pragma unroll
            for (int k = 0; k < SPW; ++k) {
                if (!on[k]) continue;
                const int senv = senv0 + k;
                const long p = (long)a.m.ptr[senv];
                const long real = (long)a.m.static_[(size_t)senv * a.m.static_rows * a.m.nR + p]; // pack.py:339
                const ClearRanges cr = clear_ranges(a.m, real);
                const float *src = a.m.dyn_in + (size_t)senv * slab;
                float *dst = a.m.dyn_out + (size_t)senv * slab;
                for (long f = lane; f < (long)slab; f += 64) {
                    float v = src[f];
                    if (in_cleared(cr, (int)f)) v = 0.f;
                    dst[f] = v;
                }
                mask_env(a.m, senv, lane, real, p);
            }
        }
        return;
    }

    // ---- placement waves (tools.py:3663-3744): their latency chain runs beside the stream --------
    __builtin_amdgcn_s_setprio(2);
    const int W = a.s.d.W, L = a.s.d.L, cells = W * L;
    const bool fresh = a.flags & TAP_T_FRESH;
    const int cell = tid % G;
    const int env = env_base + tid / G;
    const bool ev = env < B, incell = cell < cells;
    int hm = 0, cv = 0, dims[3] = {1, 1, 1};
    if (ev) {
        if (!fresh) {
            if (incell) hm = a.s.v.hm[(size_t)env * cells + cell];
            if (cell < 4) cv = a.s.v.cnt[(size_t)env * 4 + cell];
        }
        const long p = (long)a.s.ptr[env];
        for (int k = 0; k < D; ++k) // model.py:404-412
            dims[k] = (int)a.s.static_[((size_t)env * a.s.static_rows + 1 + k) * a.s.nR + p];
    }
    const int gl0 = lane - cell;
    Counters cnt = {__shfl(cv, gl0), __shfl(cv, gl0 + 1), __shfl(cv, gl0 + 2), __shfl(cv, gl0 + 3)};
    const int bx = dims[0], by = D == 3 ? dims[1] : 1, bz = dims[D - 1];
    int err = 0;
    bool do_step = ev;
    if (ev && cnt.count >= a.s.d.n_max) { err |= 2; do_step = false; }
    if (ev && (bx < 1 || by < 1 || bz < 1)) { err |= 4; do_step = false; }
    s_old[tid] = hm;
    tap_wave_lds_sync();
    const PlaceCfg cfg = {W, L, a.s.d.H, a.s.d.flags};
    const int step = cnt.count;
    const Placement pl = tap_place<D, G>(cfg, s_old + (tid - cell), cell, hm, cnt, err, bx, by, bz, do_step);
    err = group_or<G>(err);
    s_new[tid] = hm;
    tap_wave_lds_sync();
    const int gmax = (a.flags & TAP_T_RATIO) ? group_max<G>(incell ? hm : 0) : 0;
    if (ev) {
        if (incell) a.s.v.hm[(size_t)env * cells + cell] = hm;
        if (a.s.feature_out)
            tap_write_feature<D, G>(a.s.d.feature, W, L, s_new + (tid - cell), cell, hm,
                                    a.s.feature_out + (size_t)env * a.s.flen);
        if (cell == 0) {
            if (do_step || fresh)
                reinterpret_cast<int4 *>(a.s.v.cnt)[env] = make_int4(cnt.valid, cnt.empty, cnt.nstable, cnt.count);
            if (do_step) {
                int32_t *q = a.s.v.pos + (size_t)step * D * B + env;
                q[0] = pl.x;
                if (D == 3) { q[B] = pl.y; q[2 * (size_t)B] = pl.z; } else q[B] = pl.z;
                a.s.v.stable[(size_t)step * B + env] = (uint8_t)pl.stab;
            }
            if (fresh) a.s.v.err[env] = err;
            else if (err) a.s.v.err[env] |= err;
            if (a.flags & TAP_T_RATIO) { // tools.py:3887-3966 on the state just written
                double C = 0.0, P = 0.0, S = 0.0;
                if (cnt.count != 0) {
                    C = (double)cnt.valid / (double)((long long)W * L * gmax);
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

template <int D, int G, int SW>
static int launch_transition_v(tap_ctx *ctx, const TransArgs &a, hipStream_t st)
{
    constexpr int EPB = TransGeom<G, SW>::EPB, THREADS = TransGeom<G, SW>::THREADS;
    const int grid = (a.s.d.B + EPB - 1) / EPB;
    if (grid == 0) return TAP_OK;
    if (mask_fast_path_ok(a.m)) {
        const size_t lds = (size_t)EPB * 3 * a.m.nR * sizeof(float);
        hipLaunchKernelGGL((k_transition<D, G, true, SW>), dim3(grid), dim3(THREADS), lds, st, a);
    } else {
        hipLaunchKernelGGL((k_transition<D, G, false, SW>), dim3(grid), dim3(THREADS), 0, st, a);
    }
    TAP_LAUNCH_CHECK(ctx, "k_transition");
    return TAP_OK;
}

// TAP_TR_VARIANT (tuning knob, read once): 0 = 4 stream waves x 2 slabs (default);
// 2 = one stream wave per slab.
static int transition_variant()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("TAP_TR_VARIANT"); v = e ? atoi(e) : 0; }
    return v;
}

template <int D, int G> static int launch_transition(tap_ctx *ctx, const TransArgs &a, hipStream_t st)
{
    if (transition_variant() == 2) return launch_transition_v<D, G, 8>(ctx, a, st);
    return launch_transition_v<D, G, 4>(ctx, a, st);
}

extern "C" int tap_transition(tap_ctx *ctx, const tap_env_desc *d, void *state, int n, int R, int rows,
                              int update_rows, const float *dyn_in, const float *static_,
                              int static_rows, const int64_t *ptr, const float *mask_in,
                              const float *colsum_in, float *dyn_out, float *colsum_out,
                              float *current_out, float *mask_out, float *feature_out,
                              float *ratio_out, int flags, void *stream)
{
    int rc = tap_desc_validate(ctx, d);
    if (rc) return rc;
    if (d->strategy != TAP_LB_GREEDY)
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "tap_transition implements LB_GREEDY; step MACS/MUL with tap_mask_step + tap_env_step_gather");
    if (!state || !dyn_in || !static_ || !ptr || !mask_in || !colsum_in || !dyn_out || !colsum_out ||
        !current_out || !mask_out || n < 1 || R < 1 || rows < 1 || static_rows < 1 + d->D ||
        update_rows < 0 || update_rows > 3 || ((flags & TAP_T_RATIO) && !ratio_out))
        return tap_fail(ctx, TAP_E_INVALID, "bad transition arguments");
    if (dyn_in == dyn_out) return tap_fail(ctx, TAP_E_INVALID, "transition is out of place (pack.py:370)");
    TransArgs a = {};
    a.s.d = *d;
    tap_env_layout(d, state, &a.s.v);
    a.s.static_ = static_; a.s.static_rows = static_rows; a.s.nR = n * R; a.s.ptr = ptr;
    a.s.feature_out = feature_out; a.s.flen = tap_env_feature_len(d);
    a.m = MaskArgs{d->B, n, R, n * R, rows, update_rows, static_rows, dyn_in, dyn_out, static_, ptr,
                   mask_in, colsum_in, colsum_out, current_out, mask_out};
    a.flags = flags;
    a.ratio_out = ratio_out;
    const int Gs = tap_group_size(d);
    if (d->D == 2) {
        if (Gs == 8) return launch_transition<2, 8>(ctx, a, (hipStream_t)stream);
        if (Gs == 16) return launch_transition<2, 16>(ctx, a, (hipStream_t)stream);
        if (Gs == 32) return launch_transition<2, 32>(ctx, a, (hipStream_t)stream);
        return launch_transition<2, 64>(ctx, a, (hipStream_t)stream);
    }
    if (Gs == 8) return launch_transition<3, 8>(ctx, a, (hipStream_t)stream);
    if (Gs == 16) return launch_transition<3, 16>(ctx, a, (hipStream_t)stream);
    if (Gs == 32) return launch_transition<3, 32>(ctx, a, (hipStream_t)stream);
    return launch_transition<3, 64>(ctx, a, (hipStream_t)stream);
}
