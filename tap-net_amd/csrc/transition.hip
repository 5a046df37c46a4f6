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
// This file: the LB_GREEDY kernel and the tap_transition* entry points; the MACS / MUL kernels of the same shape live in
// transition_macs.hip (tap_transition_macs_launch), so that the two halves compile side by side.
#include <cstdlib>

#include "tap_common.h"
#include "tap_macs.h"
#include "tap_macs3.h"
#include "tap_masks.h"
#include "tap_place.h"
#include "tap_transition.h"
#include "tap_waves.h"

template <int D, int G, int NC, int SW, int MODE>
__global__ void __launch_bounds__((TransGeom<G, SW>::THREADS)) k_transition(TransArgs a)
{
    using Geo = TransGeom<G, SW>;
    constexpr int EPB = Geo::EPB, SPW = Geo::SPW, ENV_WAVES = Geo::ENV_WAVES;
    extern __shared__ float trans_lds[];
    __shared__ int s_old[64 * ENV_WAVES];
    __shared__ int s_new[64 * ENV_WAVES];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int env_base = blockIdx.x * EPB;

    if (wave >= ENV_WAVES) {
        trans_stream_wave<SPW, NC, MODE>(a.m, env_base + (wave - ENV_WAVES) * SPW, lane,
                                     trans_lds + (size_t)(wave - ENV_WAVES) * SPW * 3 * a.m.nR);
        return;
    }

    // ---- placement waves (tools.py:3663-3744): their latency chain runs beside the stream --------
    __builtin_amdgcn_s_setprio(2);
    const int cell = tid % G;
    tap_lb_place_wave<D, G>(a.s, a.flags, a.ratio_out, env_base + tid / G, cell, lane,
                            s_old + (tid - cell), s_new + (tid - cell));
}

int tap_transition_macs_launch(tap_ctx *ctx, const tap_env_desc *d, const TransArgs &a, hipStream_t st);   // transition_macs.hip
int tap_macs_validate(tap_ctx *ctx, const tap_env_desc &d);                                                    // macs.hip

template <int D, int G, int SW>
static int launch_transition_v(tap_ctx *ctx, const TransArgs &a, hipStream_t st)
{
    constexpr int EPB = TransGeom<G, SW>::EPB, THREADS = TransGeom<G, SW>::THREADS;
    const int grid = (a.s.d.B + EPB - 1) / EPB;
    if (grid == 0) return TAP_OK;
    const size_t lds = (size_t)EPB * 3 * a.m.nR * sizeof(float);
    const int mode = a.m.bits_in ? 1 : mask_builds_bits(a.m) ? 2 : 0;
#define TAP_LAUNCH_T(NC_, M_, LDS_) hipLaunchKernelGGL((k_transition<D, G, NC_, SW, M_>), dim3(grid), dim3(THREADS), LDS_, st, a)
#define TAP_LAUNCH_M(NC_, LDS_) do { if (mode == 1) TAP_LAUNCH_T(NC_, 1, LDS_); else if (mode == 2) TAP_LAUNCH_T(NC_, 2, LDS_); else TAP_LAUNCH_T(NC_, 0, LDS_); } while (0)
    switch (mask_fast_path_cols(a.m)) {
    case 1: TAP_LAUNCH_M(1, lds); break;
    case 2: TAP_LAUNCH_M(2, lds); break;
    case 4: TAP_LAUNCH_M(4, lds); break;
    default: TAP_LAUNCH_T(0, 0, 0); break;
    }
#undef TAP_LAUNCH_M
#undef TAP_LAUNCH_T
    TAP_LAUNCH_CHECK(ctx, "k_transition");
    return TAP_OK;
}

template <int D, int G> static int launch_transition(tap_ctx *ctx, const TransArgs &a, hipStream_t st)
{
    return launch_transition_v<D, G, 4>(ctx, a, st); // 4 stream waves: best of 1/2/4/8 for both forms of the update
}

static int transition_dispatch(tap_ctx *ctx, const tap_env_desc *d, const TransArgs &a, void *stream);

// Shapes and strategies whose placement is one THREAD per container (legacy 'LB', LB_GREEDY above 64 cells or a 3D
// side above 8) or the wide MACS 2D form (17 .. 64 columns): no single kernel carries both halves of the step.  (For
// wide MACS one was built and measured in round 3 -- tap_macs_wide_wave as the placement waves of a k_transition_macs
// look-alike: 121 against 97 us per step at W = 20, n = 12, B = 8192, equal at W = 40: the placement is ~100 us of
// register-heavy work, the stream waves hide nothing and their wave slots cost a quarter of the resident envs.)
// The tap_transition* entry points then run the same step as its two launches (precedence update, placement) plus
// reset / calc_ratio where the flags ask for them, so a caller drives every shape through one entry point.
static bool transition_single_kernel(const tap_env_desc *d, int nR)
{
    if (d->strategy == TAP_LB || tap_is_big(d) || (d->strategy == TAP_MACS && d->D == 2 && d->W > 16)) return false;
    if (d->strategy == TAP_MACS) {
        // the MACS placement keeps its lists in LDS: a container too tall (or an episode too long) for one workgroup's
        // 64 KB next to the stream tiles takes the two launches as well (the stand-alone step runs fewer envs per workgroup)
        const int G = tap_group_size(d), epb = G == 64 ? 4 : 8;
        const size_t words = d->D == 3 ? (size_t)macs3_group_words(G, d->n_max, d->H)
                                       : (size_t)macs_group_words(d->W <= 8 ? 8 : 16, d->H, d->n_max, d->W);
        if ((size_t)epb * 3 * nR * sizeof(float) + (size_t)epb * words * sizeof(int) > 64 * 1024) return false;
    }
    return true;
}

static int transition_tail(tap_ctx *ctx, const tap_env_desc *d, void *state, const TransArgs &a, void *stream)
{
    int rc = tap_env_step_gather(ctx, d, state, a.s.static_, a.s.static_rows, a.s.nR, a.s.ptr, nullptr, a.s.feature_out, stream);
    if (rc == TAP_OK && (a.flags & TAP_T_RATIO)) rc = tap_env_ratio(ctx, d, state, a.ratio_out, nullptr, nullptr, stream);
    return rc;
}

static int transition_common(tap_ctx *ctx, const tap_env_desc *d, void *state, int n, int R, int rows,
                             int update_rows, const float *static_, int static_rows, const int64_t *ptr,
                             const float *mask_in, float *current_out, float *mask_out, float *feature_out,
                             float *ratio_out, int flags, TransArgs &a)
{
    int rc = tap_desc_validate(ctx, d);
    if (rc) return rc;
    if (d->B == 0) return TAP_OK; // an empty batch has no buffers to check
    if (d->strategy == TAP_MACS && (rc = tap_macs_validate(ctx, *d)) != TAP_OK) return rc;
    if (!state || !static_ || !ptr || !mask_in || !current_out || !mask_out || n < 1 || R < 1 || rows < 1 ||
        static_rows < 1 + d->D || update_rows < 0 || update_rows > 3 || ((flags & TAP_T_RATIO) && !ratio_out))
        return tap_fail(ctx, TAP_E_INVALID, "bad transition arguments");
    a.s.d = *d;
    tap_env_layout(d, state, &a.s.v);
    a.s.static_ = static_; a.s.static_rows = static_rows; a.s.nR = n * R; a.s.ptr = ptr;
    a.s.feature_out = feature_out; a.s.flen = tap_env_feature_len(d);
    a.s.lut = ctx ? ctx->stab_lut : nullptr;
    a.flags = flags;
    a.ratio_out = ratio_out;
    return TAP_OK;
}

extern "C" int tap_transition(tap_ctx *ctx, const tap_env_desc *d, void *state, int n, int R, int rows,
                              int update_rows, const float *dyn_in, const float *static_,
                              int static_rows, const int64_t *ptr, const float *mask_in,
                              const float *colsum_in, float *dyn_out, float *colsum_out,
                              float *current_out, float *mask_out, float *feature_out,
                              float *ratio_out, int flags, void *stream)
{
    if (d && d->B == 0) return tap_desc_validate(ctx, d); // an empty batch has no buffers to check
    TransArgs a = {};
    int rc = transition_common(ctx, d, state, n, R, rows, update_rows, static_, static_rows, ptr, mask_in,
                               current_out, mask_out, feature_out, ratio_out, flags, a);
    if (rc) return rc;
    if (!dyn_in || !colsum_in || !dyn_out || !colsum_out)
        return tap_fail(ctx, TAP_E_INVALID, "bad transition arguments");
    if (dyn_in == dyn_out) return tap_fail(ctx, TAP_E_INVALID, "transition is out of place (pack.py:370)");
    if (!transition_single_kernel(d, n * R)) {
        if ((flags & TAP_T_FRESH) && (rc = tap_env_reset(ctx, d, state, stream)) != TAP_OK) return rc;
        rc = tap_mask_step(ctx, d->B, n, R, rows, update_rows, dyn_in, static_, static_rows, ptr, mask_in, colsum_in, dyn_out,
                           colsum_out, current_out, mask_out, stream);
        return rc ? rc : transition_tail(ctx, d, state, a, stream);
    }
    a.m = mask_finish(MaskArgs{d->B, n, R, n * R, rows, update_rows, static_rows, dyn_in, dyn_out, static_, ptr,
                   mask_in, colsum_in, colsum_out, current_out, mask_out, nullptr, nullptr});
    return transition_dispatch(ctx, d, a, stream);
}

extern "C" int tap_transition_bits(tap_ctx *ctx, const tap_env_desc *d, void *state, int n, int R, int rows,
                                   int update_rows, const unsigned long long *bits_in, const float *static_,
                                   int static_rows, const int64_t *ptr, const float *mask_in,
                                   unsigned long long *bits_out, float *dyn_out, float *current_out,
                                   float *mask_out, float *feature_out, float *ratio_out, int flags,
                                   void *stream)
{
    if (d && d->B == 0) return tap_desc_validate(ctx, d); // an empty batch has no buffers to check
    TransArgs a = {};
    int rc = transition_common(ctx, d, state, n, R, rows, update_rows, static_, static_rows, ptr, mask_in,
                               current_out, mask_out, feature_out, ratio_out, flags, a);
    if (rc) return rc;
    if (!bits_in || !bits_out || bits_in == bits_out)
        return tap_fail(ctx, TAP_E_INVALID, "bad transition_bits arguments");
    if (!transition_single_kernel(d, n * R) || rows > 64) {      // the fused kernels carry the one-word shadow only
        if ((flags & TAP_T_FRESH) && (rc = tap_env_reset(ctx, d, state, stream)) != TAP_OK) return rc;
        rc = tap_mask_step_bits(ctx, d->B, n, R, rows, update_rows, bits_in, static_, static_rows, ptr, mask_in, bits_out,
                                dyn_out, current_out, mask_out, stream);
        return rc ? rc : transition_tail(ctx, d, state, a, stream);
    }
    a.m = mask_finish(MaskArgs{d->B, n, R, n * R, rows, update_rows, static_rows, nullptr, dyn_out, static_, ptr,
                   mask_in, nullptr, nullptr, current_out, mask_out, bits_in, bits_out});
    if (!mask_bits_ok(a.m))
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "bit shadow needs nR %% 4 == 0, nR <= 256, rows <= 128, 16-byte aligned buffers");
    return transition_dispatch(ctx, d, a, stream);
}

extern "C" int tap_transition_first(tap_ctx *ctx, const tap_env_desc *d, void *state, int n, int R, int rows,
                                    int update_rows, const float *dyn_in, const float *static_, int static_rows,
                                    const int64_t *ptr, const float *mask_in, unsigned long long *bits_out,
                                    float *dyn_out, float *current_out, float *mask_out, float *feature_out,
                                    float *ratio_out, int32_t *nonbinary_out, int flags, void *stream)
{
    if (d && d->B == 0) return tap_desc_validate(ctx, d); // an empty batch has no buffers to check
    TransArgs a = {};
    int rc = transition_common(ctx, d, state, n, R, rows, update_rows, static_, static_rows, ptr, mask_in,
                               current_out, mask_out, feature_out, ratio_out, flags, a);
    if (rc) return rc;
    if (!dyn_in || !bits_out || dyn_in == dyn_out)
        return tap_fail(ctx, TAP_E_INVALID, "bad transition_first arguments");
    if (!transition_single_kernel(d, n * R) || rows > 64) {
        if ((flags & TAP_T_FRESH) && (rc = tap_env_reset(ctx, d, state, stream)) != TAP_OK) return rc;
        rc = tap_mask_step_first(ctx, d->B, n, R, rows, update_rows, dyn_in, static_, static_rows, ptr, mask_in, bits_out,
                                 dyn_out, current_out, mask_out, nonbinary_out, stream);
        return rc ? rc : transition_tail(ctx, d, state, a, stream);
    }
    a.m = mask_finish(MaskArgs{d->B, n, R, n * R, rows, update_rows, static_rows, dyn_in, dyn_out, static_, ptr,
                   mask_in, nullptr, nullptr, current_out, mask_out, nullptr, bits_out, nonbinary_out});
    if (!mask_bits_ok(a.m))
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "bit shadow needs nR %% 4 == 0, nR <= 256, rows <= 128, 16-byte aligned buffers");
    return transition_dispatch(ctx, d, a, stream);
}

static int transition_dispatch(tap_ctx *ctx, const tap_env_desc *d, const TransArgs &a, void *stream)
{
    const int Gs = tap_group_size(d);
    if (d->strategy == TAP_MACS) return tap_transition_macs_launch(ctx, d, a, (hipStream_t)stream);
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
