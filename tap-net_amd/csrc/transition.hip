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
#include <new>

#include "tap_common.h"
#include "tap_macs.h"
#include "tap_macs3.h"
#include "tap_masks.h"
#include "tap_place.h"
#include "tap_transition.h"
#include "tap_waves.h"

// (TAP_MASK_HOT_PARAMS, tap_masks.h: the load addresses of the stream waves -- and of the placement waves' gather, which
//  reads the same `ptr` and `static` -- arrive in SGPRs with the wave)
template <int D, int G, int NC, int SW, int MODE>
__global__ void __launch_bounds__((TransGeom<G, SW>::THREADS)) k_transition(TAP_MASK_HOT_PARAMS, TransArgs a)
{
    using Geo = TransGeom<G, SW>;
    constexpr int EPB = Geo::EPB, SPW = Geo::SPW, ENV_WAVES = Geo::ENV_WAVES;
    extern __shared__ float trans_lds[];
    __shared__ int s_old[64 * ENV_WAVES];
    __shared__ int s_new[64 * ENV_WAVES];
    // The wave's index in a SCALAR register (two-slab stream waves: 2D windows): everything a stream wave derives from it --
    // its envs, every load and store address -- is then scalar arithmetic + lane offsets instead of 64-bit vector
    // multiplies in front of the first load.  Round 6, same session (scripts/ab_transition.sh): c2 1 380 -> 1 440 M
    // env-steps/s; c3 (one slab per wave) 545 -> 541 M, the MACS steps flat, the rolling step 530 -> 491 M -- so only here,
    // and only for SPW > 1.  -DTAP_VWAVE: the index as the compiler sees it (a VGPR) everywhere (A/B builds).
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef TAP_VWAVE
    const int wave = tid >> 6;
#else
    const int wave = TransGeom<G, SW>::SPW > 1 ? __builtin_amdgcn_readfirstlane(tid >> 6) : tid >> 6;
#endif
    const int env_base = blockIdx.x * EPB;

#if defined(TAP_PROF) || defined(TAP_PROF_SWITCH)
    // decomposition switches (scripts/decompose_step.py): run only one kind of wave, or none ("empty kernel of the geometry")
    if (wave >= ENV_WAVES ? (a.flags & TAP_T_PROF_NOSTREAM) : (a.flags & TAP_T_PROF_NOPLACE)) return;
#endif
    if (wave >= ENV_WAVES) {
#ifdef TAP_STREAM_PRIO                                       // A/B builds: the stream waves' issue priority (default 0)
        __builtin_amdgcn_s_setprio(TAP_STREAM_PRIO);
#endif
        const MaskArgs m = tap_mask_hot(a.m, (MODE & 3) == 1, TAP_MASK_HOT_NAMES);
        trans_stream_wave<SPW, NC, MODE>(m, env_base + (wave - ENV_WAVES) * SPW, lane,
                                     trans_lds + (size_t)(wave - ENV_WAVES) * SPW * 3 * h_nR);
        return;
    }

    // ---- placement waves (tools.py:3663-3744): their latency chain runs beside the stream --------
#ifdef TAP_PLACE_PRIO                                        // A/B builds
    __builtin_amdgcn_s_setprio(TAP_PLACE_PRIO);
#else
    __builtin_amdgcn_s_setprio(2);
#endif
    const int cell = tid % G;
    TL_STAMP(0);
    StepArgs sa = a.s;       // the gather's source is the update's: ptr, static and their sizes are already in SGPRs
    sa.ptr = h_ptr; sa.static_ = h_static; sa.nR = h_nR; sa.static_rows = h_static_rows; sa.d.B = h_B;
    tap_lb_place_wave<D, G>(sa, a.flags, a.ratio_out, env_base + tid / G, cell, lane,
                            s_old + (tid - cell), s_new + (tid - cell));
    TL_WAIT_VM();
    TL_STAMP(3);
}

#ifdef TAP_PROF
extern "C" int tap_prof_read_timeline(unsigned long long *out, int clear)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(tap_prof_tl), sizeof(unsigned long long) * TAP_PROF_WGS * TAP_PROF_WAVES * 4);
    if (clear) {
        static unsigned long long zeros[TAP_PROF_WGS * TAP_PROF_WAVES * 4];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(tap_prof_tl), zeros, sizeof(zeros));
    }
    return 0;
}
#endif

// (Round 4, measured and removed: ONE LANE per container for the 2D placement inside this kernel -- the corners walked
//  one after the other with the height-map in registers, no LDS, no cross-lane step -- to cut the placement's vector
//  instructions for large batches.  SQ_INSTS_VALU per launch at c2 went UP, 1.94 M -> 2.27 M (the corner and footprint
//  loops run predicated over 8 x 8 register cells on every lane), the step from 6.86 to 7.48 us at B = 8192 (the serial
//  chain of up to five scored corners becomes the launch's tail), and at B = 1 M nothing moved (877 against 898 M
//  env-steps/s): the saturated regime is not bound by vector issue -- 248 M vector instructions per 1.16 ms launch are
//  35 % of the SIMDs' issue slots.)
int tap_transition_macs_launch(tap_ctx *ctx, const tap_env_desc *d, const TransArgs &a, hipStream_t st);   // transition_macs.hip
// big.hip: LB_GREEDY containers above 64 cells -- one wavefront per container + stream waves, on the bit shadow
bool tap_big_transition_ok(const tap_ctx *ctx, const tap_env_desc *d, int nR);
int tap_big_transition(tap_ctx *ctx, const tap_env_desc *d, const TransArgs &a, hipStream_t st);
// macs_big.hip / macs3_big.hip: the wave-per-container MACS placements with the stream waves beside them (mask update +
// placement in one launch; a fresh container and calc_ratio stay launches of their own at the two ends of an episode)
bool tap_macs_wave_transition_ok(const tap_ctx *ctx, const tap_env_desc *d, int nR);
int tap_macs_wave_transition(tap_ctx *ctx, const tap_env_desc *d, const TransArgs &a, hipStream_t st);
bool tap_macs3_wave_transition_ok(const tap_ctx *ctx, const tap_env_desc *d, int nR);
int tap_macs3_wave_transition(tap_ctx *ctx, const tap_env_desc *d, const TransArgs &a, hipStream_t st);
int tap_macs2d_wave_from();                                                                                 // macs.hip

// 0: no wave-per-container fused step; 1: big.hip (LB_GREEDY, handles fresh / ratio itself); 2 / 3: MACS 2D / 3D
static int transition_wave_kind(const tap_ctx *ctx, const tap_env_desc *d, int nR, int rows)
{
    if (rows > 64) return 0;                                                      // the fused kernels carry the one-word shadow only
    if (tap_is_big(d)) return tap_big_transition_ok(ctx, d, nR) ? 1 : 0;
    if (d->strategy != TAP_MACS) return 0;
    if (d->D == 2) return (d->W > 16 && d->W >= tap_macs2d_wave_from() && tap_macs_wave_transition_ok(ctx, d, nR)) ? 2 : 0;
    return (tap_is_big_macs3(d) && tap_macs3_wave_transition_ok(ctx, d, nR)) ? 3 : 0;
}

// the step of a wave-per-container shape on the bit shadow: a.m is filled in
static int transition_wave_launch(tap_ctx *ctx, const tap_env_desc *d, void *state, const TransArgs &a, int kind, void *stream)
{
    if (kind == 1) return tap_big_transition(ctx, d, a, (hipStream_t)stream);
    int rc = TAP_OK;
    if ((a.flags & TAP_T_FRESH) && (rc = tap_env_reset(ctx, d, state, stream)) != TAP_OK) return rc;
    rc = kind == 2 ? tap_macs_wave_transition(ctx, d, a, (hipStream_t)stream) : tap_macs3_wave_transition(ctx, d, a, (hipStream_t)stream);
    if (rc == TAP_OK && (a.flags & TAP_T_RATIO)) rc = tap_env_ratio(ctx, d, state, a.ratio_out, nullptr, nullptr, stream);
    return rc;
}
int tap_macs_validate(tap_ctx *ctx, const tap_env_desc &d);                                                    // macs.hip

template <int D, int G, int SW>
static int launch_transition_v(tap_ctx *ctx, const TransArgs &a, hipStream_t st)
{
    constexpr int EPB = TransGeom<G, SW>::EPB, THREADS = TransGeom<G, SW>::THREADS;
    const int grid = (a.s.d.B + EPB - 1) / EPB;
    if (grid == 0) return TAP_OK;
    const size_t lds = (size_t)EPB * 3 * a.m.nR * sizeof(float);
    const int mode = a.m.bits_in ? 1 : mask_builds_bits(a.m) ? 2 : 0;
#define TAP_LAUNCH_K(NC_, M_, LDS_) hipLaunchKernelGGL((k_transition<D, G, NC_, SW, M_>), dim3(grid), dim3(THREADS), LDS_, st, TAP_MASK_HOT_ARGS(a.m), a)
    // the reference's own window (n = 10: rows = 30, nR = 20 / 60) on the bit shadow runs the instantiation with its shape
    // compiled in (tap_transition.h: TAP_MODE_C4_*)
    const bool shaped = tap_mode_shape_ok(a.m, D);
    // ... and, for a step on a shadow the caller hands in, without the code for absent inputs and idle slabs when there are none
#ifdef TAP_NO_FULL                                            // A/B builds
    const bool full = false;
#else
    const bool full = mode == 1 && a.m.ptr && a.m.static_ && a.m.mask_in && a.s.d.B % EPB == 0;
#endif
#define TAP_LAUNCH_T(NC_, M_, LDS_) do { if constexpr ((NC_) == 1 && ((M_) & 3) == 1) { \
            if (shaped && full) TAP_LAUNCH_K(NC_, ((M_) | tap_mode_shape(D) | TAP_MODE_FULL), LDS_); \
            else if (shaped) TAP_LAUNCH_K(NC_, ((M_) | tap_mode_shape(D)), LDS_); else TAP_LAUNCH_K(NC_, M_, LDS_); } \
        else if constexpr ((NC_) == 1 && ((M_) & 3) != 0) { if (shaped) TAP_LAUNCH_K(NC_, ((M_) | tap_mode_shape(D)), LDS_); else TAP_LAUNCH_K(NC_, M_, LDS_); } \
        else TAP_LAUNCH_K(NC_, M_, LDS_); } while (0)
    // 2D windows (nR = 2n columns: five store instructions per run at c2) take the run-of-rows expansion while the stores
    // are write-through; 3D windows and every launch beyond the write-through limit keep the slab-by-slab loops
    const bool merged = D == 2 && a.m.wt != 0;
    // the caller's dyn_out already holds the previous step's tensor (a stepper on ONE dyn buffer): only the cleared rows are written
    const bool inpl = mode == 1 && a.m.inplace && a.m.dyn_out;
#define TAP_LAUNCH_M(NC_, LDS_) do { if (inpl) TAP_LAUNCH_T(NC_, (1 | TAP_MODE_INPLACE), LDS_); \
        else if (mode == 1) { if (D == 2 && merged) TAP_LAUNCH_T(NC_, (D == 2 ? 5 : 1), LDS_); else TAP_LAUNCH_T(NC_, 1, LDS_); } \
        else if (mode == 2) { if (D == 2 && merged) TAP_LAUNCH_T(NC_, (D == 2 ? 6 : 2), LDS_); else TAP_LAUNCH_T(NC_, 2, LDS_); } else TAP_LAUNCH_T(NC_, 0, LDS_); } while (0)
    switch (mask_fast_path_cols(a.m)) {
    case 1: TAP_LAUNCH_M(1, lds); break;
    case 2: TAP_LAUNCH_M(2, lds); break;
    case 4: TAP_LAUNCH_M(4, lds); break;
    default: TAP_LAUNCH_T(0, 0, 0); break;
    }
#undef TAP_LAUNCH_M
#undef TAP_LAUNCH_T
#undef TAP_LAUNCH_K
    (void)shaped; (void)inpl; (void)full;
    TAP_LAUNCH_CHECK(ctx, "k_transition");
    return TAP_OK;
}

template <int D, int G> static int launch_transition(tap_ctx *ctx, const TransArgs &a, hipStream_t st)
{
    // Stream waves per workgroup of 8 envs.  2D windows (2 400-byte slabs at n = 10): 4, two slabs per wave (8: c2 1 349 ->
    // 1 216 M env-steps/s, 1 903 -> 1 753 M at B = 65 536).  3D windows (7 200-byte slabs): 8, one slab per wave -- c3
    // 507 -> 543 M at B = 4 096, 556 -> 598 M at 8 192, 549 -> 574 M at 16 384, 605 -> 620 M at 65 536, equal from 262 144
    // (round 5, on the kernels with the compiled-in window shape; round 4 had measured 4 as the best for both at c2's
    // shape only).  -DTAP_TRANS_SW=n forces one value (scripts/ab_transition.sh).
#ifdef TAP_TRANS_SW
    return launch_transition_v<D, G, TAP_TRANS_SW>(ctx, a, st);
#else
    return launch_transition_v<D, G, (D == 3 ? 8 : 4)>(ctx, a, st);
#endif
}

static int transition_dispatch(tap_ctx *ctx, const tap_env_desc *d, const TransArgs &a, void *stream);

// Shapes and strategies whose placement is one THREAD per container (legacy 'LB', LB_GREEDY and MACS 3D above 64
// cells or with a 3D side above 8) or the wide MACS 2D form (17 .. 64 columns): no single kernel carries both halves of the step.  (For
// wide MACS one was built and measured in round 3 -- tap_macs_wide_wave as the placement waves of a k_transition_macs
// look-alike: 121 against 97 us per step at W = 20, n = 12, B = 8192, equal at W = 40: the placement is ~100 us of
// register-heavy work, the stream waves hide nothing and their wave slots cost a quarter of the resident envs.)
// The tap_transition* entry points then run the same step as its two launches (precedence update, placement) plus
// reset / calc_ratio where the flags ask for them, so a caller drives every shape through one entry point.
static bool transition_single_kernel(const tap_ctx *ctx, const tap_env_desc *d, int nR)
{
    if (d->strategy == TAP_LB || tap_is_big(d) || tap_is_big_macs3(d) || (d->strategy == TAP_MACS && d->D == 2 && d->W > 16)) return false;
    if (d->strategy == TAP_MACS) {
        // the MACS placement keeps its lists in LDS: a container too tall (or an episode too long) for one workgroup's
        // LDS (160 KiB on gfx950) next to the stream tiles takes the two launches as well (the stand-alone step runs
        // fewer envs per workgroup)
        const int G = tap_group_size(d), epb = G == 64 ? 4 : 8;
        const size_t words = d->D == 3 ? (size_t)macs3_group_words(G, d->n_max, d->H)
                                       : (size_t)macs_group_words(d->W <= 8 ? 8 : 16, d->H, d->n_max, d->W);
        if ((size_t)epb * 3 * nR * sizeof(float) + (size_t)epb * words * sizeof(int) > tap_lds_limit(ctx)) return false;
    }
    return true;
}

// The by-products a decoding loop wants from the step's gather (tap_common.h: StepArgs::dec_static_out / tour_out):
// the fused kernels' placement waves write them; for the shapes that run as two launches this kernel does.
struct StepAux {
    float *dec_static_out;
    int64_t *tour_out;
    int tour_stride, tour_col;
};

__global__ void __launch_bounds__(TAP_BLOCK) k_step_aux(StepArgs s)
{
    const int env = blockIdx.x * TAP_BLOCK + threadIdx.x;
    if (env >= s.d.B) return;
    bool badp;
    const long praw = (long)s.ptr[env];
    const long p = tap_col(praw, s.nR, badp);
    float fv[3] = {0.f, 0.f, 0.f};
    for (int k = 0; k < s.d.D; ++k) {
        const float v = s.static_[((size_t)env * s.static_rows + 1 + k) * s.nR + p];
        fv[k] = badp ? 0.f : v;
    }
    tap_step_aux(s, env, s.d.D, fv, praw);
}

int tap_step_aux_launch(tap_ctx *ctx, const StepArgs &s, hipStream_t st)
{
    if (!(s.dec_static_out || s.tour_out || s.picked_out) || s.d.B == 0) return TAP_OK;
    hipLaunchKernelGGL(k_step_aux, dim3((s.d.B + TAP_BLOCK - 1) / TAP_BLOCK), dim3(TAP_BLOCK), 0, st, s);
    TAP_LAUNCH_CHECK(ctx, "k_step_aux");
    return TAP_OK;
}

extern "C" int tap_transition_launches(const tap_ctx *ctx, const tap_env_desc *d, int n, int R, int rows, int on_bits)
{
    if (!d || n < 1 || R < 1 || rows < 1) return TAP_E_INVALID;
    if (on_bits && transition_wave_kind(ctx, d, n * R, rows)) return 1;
    return (transition_single_kernel(ctx, d, n * R) && (!on_bits || rows <= 64)) ? 1 : 2;
}

static int transition_tail(tap_ctx *ctx, const tap_env_desc *d, void *state, const TransArgs &a, void *stream)
{
    int rc = tap_env_step_gather(ctx, d, state, a.s.static_, a.s.static_rows, a.s.nR, a.s.ptr, nullptr, a.s.feature_out, stream);
    if (rc == TAP_OK && (a.flags & TAP_T_RATIO)) rc = tap_env_ratio(ctx, d, state, a.ratio_out, nullptr, nullptr, stream);
    return rc ? rc : tap_step_aux_launch(ctx, a.s, (hipStream_t)stream);
}

static int transition_common(tap_ctx *ctx, const tap_env_desc *d, void *state, int n, int R, int rows,
                             int update_rows, const float *static_, int static_rows, const int64_t *ptr,
                             const float *mask_in, float *current_out, float *mask_out, float *feature_out,
                             float *ratio_out, int flags, TransArgs &a, const StepAux *aux = nullptr)
{
    int rc = tap_desc_validate(ctx, d);
    if (rc) return rc;
    if (d->B == 0) return TAP_OK; // an empty batch has no buffers to check
    if (d->strategy == TAP_MACS && (rc = tap_macs_validate(ctx, *d)) != TAP_OK) return rc;
    // mask_in null = ones (the mask DRL.forward starts from, model.py:297): only the stepper's first step passes it
    if (!state || !static_ || !ptr || (!mask_in && !aux) || !current_out || !mask_out || n < 1 || R < 1 || rows < 1 ||
        static_rows < 1 + d->D || update_rows < 0 || update_rows > 3 || ((flags & TAP_T_RATIO) && !ratio_out))
        return tap_fail(ctx, TAP_E_INVALID, "bad transition arguments");
    a.s.d = *d;
    tap_env_layout(d, state, &a.s.v);
    a.s.static_ = static_; a.s.static_rows = static_rows; a.s.nR = n * R; a.s.ptr = ptr;
    a.s.feature_out = feature_out; a.s.flen = tap_env_feature_len(d);
    a.s.lut = ctx ? ctx->stab_lut : nullptr;
    a.flags = flags;
    a.ratio_out = ratio_out;
    if (aux) {
        a.s.dec_static_out = aux->dec_static_out;
        a.s.tour_out = aux->tour_out;
        a.s.tour_stride = aux->tour_stride;
        a.s.tour_col = aux->tour_col;
    }
    return TAP_OK;
}

static int transition_copy_impl(tap_ctx *ctx, const tap_env_desc *d, void *state, int n, int R, int rows,
                                int update_rows, const float *dyn_in, const float *static_,
                                int static_rows, const int64_t *ptr, const float *mask_in,
                                const float *colsum_in, float *dyn_out, float *colsum_out,
                                float *current_out, float *mask_out, float *feature_out,
                                float *ratio_out, int flags, void *stream, const StepAux *aux)
{
    if (d && d->B == 0) return tap_desc_validate(ctx, d); // an empty batch has no buffers to check
    TransArgs a = {};
    int rc = transition_common(ctx, d, state, n, R, rows, update_rows, static_, static_rows, ptr, mask_in,
                               current_out, mask_out, feature_out, ratio_out, flags, a, aux);
    if (rc) return rc;
    if (!dyn_in || !colsum_in || !dyn_out || !colsum_out)
        return tap_fail(ctx, TAP_E_INVALID, "bad transition arguments");
    if (dyn_in == dyn_out) return tap_fail(ctx, TAP_E_INVALID, "transition is out of place (pack.py:370)");
    if (!transition_single_kernel(ctx, d, n * R)) {
        if ((flags & TAP_T_FRESH) && (rc = tap_env_reset(ctx, d, state, stream)) != TAP_OK) return rc;
        rc = tap_mask_step(ctx, d->B, n, R, rows, update_rows, dyn_in, static_, static_rows, ptr, mask_in, colsum_in, dyn_out,
                           colsum_out, current_out, mask_out, stream);
        return rc ? rc : transition_tail(ctx, d, state, a, stream);
    }
    a.m = mask_finish(MaskArgs{d->B, n, R, n * R, rows, update_rows, static_rows, dyn_in, dyn_out, static_, ptr,
                   mask_in, colsum_in, colsum_out, current_out, mask_out, nullptr, nullptr});
    return transition_dispatch(ctx, d, a, stream);
}

extern "C" int tap_transition(tap_ctx *ctx, const tap_env_desc *d, void *state, int n, int R, int rows,
                              int update_rows, const float *dyn_in, const float *static_,
                              int static_rows, const int64_t *ptr, const float *mask_in,
                              const float *colsum_in, float *dyn_out, float *colsum_out,
                              float *current_out, float *mask_out, float *feature_out,
                              float *ratio_out, int flags, void *stream)
{
    if (d && d->B > 0 && !mask_in) return tap_fail(ctx, TAP_E_INVALID, "bad transition arguments");
    return transition_copy_impl(ctx, d, state, n, R, rows, update_rows, dyn_in, static_, static_rows, ptr, mask_in, colsum_in,
                                dyn_out, colsum_out, current_out, mask_out, feature_out, ratio_out, flags, stream, nullptr);
}

static int transition_bits_impl(tap_ctx *ctx, const tap_env_desc *d, void *state, int n, int R, int rows,
                                int update_rows, const unsigned long long *bits_in, const float *static_,
                                int static_rows, const int64_t *ptr, const float *mask_in,
                                unsigned long long *bits_out, float *dyn_out, float *current_out,
                                float *mask_out, float *feature_out, float *ratio_out, int flags,
                                void *stream, const StepAux *aux, int inplace = 0)
{
    // inplace: dyn_out holds update_dynamic's INPUT (the stepper's previous step wrote it there) -- the lane-per-cell fused
    // kernels then write the cleared rows only; the other paths below write the whole tensor, which is the same tensor
    if (d && d->B == 0) return tap_desc_validate(ctx, d); // an empty batch has no buffers to check
    TransArgs a = {};
    int rc = transition_common(ctx, d, state, n, R, rows, update_rows, static_, static_rows, ptr, mask_in,
                               current_out, mask_out, feature_out, ratio_out, flags, a, aux);
    if (rc) return rc;
    if (!bits_in || !bits_out || bits_in == bits_out)
        return tap_fail(ctx, TAP_E_INVALID, "bad transition_bits arguments");
    if (const int kind = transition_wave_kind(ctx, d, n * R, rows)) {
        a.m = mask_finish(MaskArgs{d->B, n, R, n * R, rows, update_rows, static_rows, nullptr, dyn_out, static_, ptr,
                       mask_in, nullptr, nullptr, current_out, mask_out, bits_in, bits_out});
        if (!mask_bits_ok(a.m))
            return tap_fail(ctx, TAP_E_UNSUPPORTED, "bit shadow needs nR %% 4 == 0, nR <= 256, rows <= 128, 16-byte aligned buffers");
        return transition_wave_launch(ctx, d, state, a, kind, stream);
    }
    if (!transition_single_kernel(ctx, d, n * R) || rows > 64) {      // the fused kernels carry the one-word shadow only
        if ((flags & TAP_T_FRESH) && (rc = tap_env_reset(ctx, d, state, stream)) != TAP_OK) return rc;
        rc = tap_mask_step_bits(ctx, d->B, n, R, rows, update_rows, bits_in, static_, static_rows, ptr, mask_in, bits_out,
                                dyn_out, current_out, mask_out, stream);
        return rc ? rc : transition_tail(ctx, d, state, a, stream);
    }
    a.m = mask_finish(MaskArgs{d->B, n, R, n * R, rows, update_rows, static_rows, nullptr, dyn_out, static_, ptr,
                   mask_in, nullptr, nullptr, current_out, mask_out, bits_in, bits_out});
    a.m.inplace = (inplace && dyn_out) ? 1 : 0;
    if (!mask_bits_ok(a.m))
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "bit shadow needs nR %% 4 == 0, nR <= 256, rows <= 128, 16-byte aligned buffers");
    return transition_dispatch(ctx, d, a, stream);
}

extern "C" int tap_transition_bits(tap_ctx *ctx, const tap_env_desc *d, void *state, int n, int R, int rows,
                                   int update_rows, const unsigned long long *bits_in, const float *static_,
                                   int static_rows, const int64_t *ptr, const float *mask_in,
                                   unsigned long long *bits_out, float *dyn_out, float *current_out,
                                   float *mask_out, float *feature_out, float *ratio_out, int flags,
                                   void *stream)
{
    return transition_bits_impl(ctx, d, state, n, R, rows, update_rows, bits_in, static_, static_rows, ptr, mask_in, bits_out,
                                dyn_out, current_out, mask_out, feature_out, ratio_out, flags, stream, nullptr);
}

static int transition_first_impl(tap_ctx *ctx, const tap_env_desc *d, void *state, int n, int R, int rows,
                                 int update_rows, const float *dyn_in, const float *static_, int static_rows,
                                 const int64_t *ptr, const float *mask_in, unsigned long long *bits_out,
                                 float *dyn_out, float *current_out, float *mask_out, float *feature_out,
                                 float *ratio_out, int32_t *nonbinary_out, int flags, void *stream, const StepAux *aux)
{
    if (d && d->B == 0) return tap_desc_validate(ctx, d); // an empty batch has no buffers to check
    TransArgs a = {};
    int rc = transition_common(ctx, d, state, n, R, rows, update_rows, static_, static_rows, ptr, mask_in,
                               current_out, mask_out, feature_out, ratio_out, flags, a, aux);
    if (rc) return rc;
    if (!dyn_in || !bits_out || dyn_in == dyn_out)
        return tap_fail(ctx, TAP_E_INVALID, "bad transition_first arguments");
    if (const int kind = transition_wave_kind(ctx, d, n * R, rows)) {
        a.m = mask_finish(MaskArgs{d->B, n, R, n * R, rows, update_rows, static_rows, dyn_in, dyn_out, static_, ptr,
                       mask_in, nullptr, nullptr, current_out, mask_out, nullptr, bits_out, nonbinary_out});
        if (!mask_bits_ok(a.m))
            return tap_fail(ctx, TAP_E_UNSUPPORTED, "bit shadow needs nR %% 4 == 0, nR <= 256, rows <= 128, 16-byte aligned buffers");
        return transition_wave_launch(ctx, d, state, a, kind, stream);
    }
    if (!transition_single_kernel(ctx, d, n * R) || rows > 64) {
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

extern "C" int tap_transition_first(tap_ctx *ctx, const tap_env_desc *d, void *state, int n, int R, int rows,
                                    int update_rows, const float *dyn_in, const float *static_, int static_rows,
                                    const int64_t *ptr, const float *mask_in, unsigned long long *bits_out,
                                    float *dyn_out, float *current_out, float *mask_out, float *feature_out,
                                    float *ratio_out, int32_t *nonbinary_out, int flags, void *stream)
{
    return transition_first_impl(ctx, d, state, n, R, rows, update_rows, dyn_in, static_, static_rows, ptr, mask_in, bits_out,
                                 dyn_out, current_out, mask_out, feature_out, ratio_out, nonbinary_out, flags, stream, nullptr);
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

// ---- tap_stepper: the step object of a decoding loop (tapenv.h) --------------------------------------------------
// Host-side only: the buffers stay the caller's; the stepper remembers them, alternates the two phases and tells
// the fused launch which by-products to write.  A step is then one C call with two arguments.
struct tap_stepper {
    tap_ctx *ctx;
    tap_env_desc d;
    void *state;
    int n, R, rows, update_rows, static_rows, steps;
    tap_stepper_buffers b;
    const float *static_;
    const float *dyn_in;                // step 0 builds the shadow from this fp32 tensor ...
    const unsigned long long *bits0;    // ... or reads this one (begin's own launch, or the caller's)
    const float *mask0;                 // the mask step 0 starts from (null = ones)
    int k;                              // index of the next step
    int keep;                           // TAP_SB_CONTINUE: step 0 does not start from a fresh container
    int copy_form;                      // the window has no bit shadow: `dynamic` is carried as the fp32 tensor + column sums
    int inplace;                        // dyn[0] == dyn[1]: ONE fp32 tensor, updated in place from the second step on
};

struct DeviceGuard { // the launches must be issued with the context's device current (callers may sit on another one)
    int prev, want;
    explicit DeviceGuard(int dev) : prev(-1), want(dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != want) (void)hipSetDevice(want);
    }
    ~DeviceGuard() { if (prev >= 0 && prev != want) (void)hipSetDevice(prev); }
};

extern "C" int tap_stepper_create(tap_ctx *ctx, const tap_env_desc *d, void *state, int n, int R, int rows,
                                  int update_rows, int static_rows, int steps, const tap_stepper_buffers *buf,
                                  tap_stepper **out)
{
    if (!ctx || !d || !buf || !out) return TAP_E_INVALID;
    int rc = tap_desc_validate(ctx, d);
    if (rc) return rc;
    if (n < 1 || R < 1 || rows < 1 || update_rows < 0 || update_rows > 3 || static_rows < 1 + d->D || steps < 1)
        return tap_fail(ctx, TAP_E_INVALID, "bad stepper arguments");
    if (d->B > 0) {                                  // an empty batch has no buffers to check
        if (!state) return tap_fail(ctx, TAP_E_INVALID, "bad stepper arguments");
        const bool shadow = n * R % 4 == 0 && n * R <= 256 && rows <= 128;
        for (int w = 0; w < 2; ++w)
            if ((shadow && !buf->bits[w]) || !buf->current[w] || !buf->mask[w])
                return tap_fail(ctx, TAP_E_INVALID, "stepper needs both phases of bits / current / mask");
        if (!shadow) {
            // a window beyond the bit shadow (nR % 4 != 0, nR > 256 or rows > 128): the step is tap_transition's fp32-copy
            // form on the column-sum shadow -- both phases of dyn and of colsum (B, 3, nR) are required
            if (!buf->dyn[0] || !buf->dyn[1] || !buf->colsum[0] || !buf->colsum[1] || buf->colsum[0] == buf->colsum[1])
                return tap_fail(ctx, TAP_E_INVALID, "a stepper for a window without a bit shadow needs both phases of dyn and colsum");
        }
        // dyn: both phases, or neither -- a caller whose encoder consumes the bit shadow skips the fp32 expansion of
        // update_dynamic's result (78 % of a c2 step's bytes); masks, placements and ratio do not depend on it
        if ((buf->dyn[0] == nullptr) != (buf->dyn[1] == nullptr))
            return tap_fail(ctx, TAP_E_INVALID, "stepper needs both phases of dyn, or neither (no fp32 expansion)");
        // dyn[0] == dyn[1] (windows with a bit shadow): ONE fp32 tensor for the whole episode.  The step never reads it (the
        // shadow carries `dynamic`), and update_dynamic's result differs from its input in the chosen rows only, so from
        // the second step on a step writes those rows' zeros instead of the whole tensor (the reference's clone, pack.py:368,
        // exists for autograd: a loop under no_grad -- validation, serving -- has no use for the previous tensors)
        if ((shadow && buf->bits[0] == buf->bits[1]) || (!shadow && buf->dyn[0] == buf->dyn[1]) || buf->current[0] == buf->current[1] ||
            buf->mask[0] == buf->mask[1] || !buf->ratio)
            return tap_fail(ctx, TAP_E_INVALID, "stepper phases must be distinct buffers (dyn may be one buffer on the bit shadow) and ratio is required");
    }
    const int nR = n * R;
    if (buf->tour_stride < 0 || buf->tour_col0 < 0 || (buf->tour_stride > 0 && buf->tour_col0 + steps > buf->tour_stride))
        return tap_fail(ctx, TAP_E_INVALID, "stepper tour columns [col0, col0 + steps) must fit tour_stride");
    tap_stepper *s = new (std::nothrow) tap_stepper();
    if (!s) return tap_fail(ctx, TAP_E_INVALID, "out of host memory");
    s->ctx = ctx; s->d = *d; s->state = state;
    s->n = n; s->R = R; s->rows = rows; s->update_rows = update_rows; s->static_rows = static_rows; s->steps = steps;
    s->b = *buf;
    s->static_ = nullptr; s->dyn_in = nullptr; s->bits0 = nullptr; s->mask0 = nullptr; s->k = 0; s->keep = 0;
    s->copy_form = (nR % 4 != 0 || nR > 256 || rows > 128) ? 1 : 0;
    s->inplace = (!s->copy_form && buf->dyn[0] && buf->dyn[0] == buf->dyn[1]) ? 1 : 0;
    *out = s;
    return TAP_OK;
}

extern "C" void tap_stepper_destroy(tap_stepper *s) { delete s; }

extern "C" int tap_stepper_begin(tap_stepper *s, const float *static_, const float *dyn_in, int flags, void *stream)
{
    if (!s) return TAP_E_INVALID;
    if ((!static_ || !dyn_in) && s->d.B > 0) return tap_fail(s->ctx, TAP_E_INVALID, "stepper_begin needs static and dynamic");
    if (s->d.B == 0) static_ = reinterpret_cast<const float *>(s); // any non-null token: "begun"
    s->static_ = static_;
    s->dyn_in = dyn_in;
    s->bits0 = nullptr;
    s->mask0 = nullptr;
    s->k = 0;
    s->keep = (flags & TAP_SB_CONTINUE) != 0;
    if (s->copy_form && s->d.B > 0) {
        // no bit shadow: the column sums of the fresh tensor and the masks DRL.forward starts from (model.py:297-307;
        // mask = ones) into phase 1, which step 0 reads -- two small launches, with or without TAP_SB_INITIAL_MASK
        DeviceGuard g(s->ctx->device);
        int rc = tap_dyn_colsum(s->ctx, s->d.B, s->n, s->n * s->R, s->rows, dyn_in, s->b.colsum[1], stream);
        // the incremental column sums are exact for 0/1 tensors only: count the other elements like the shadow builders do
        if (rc == TAP_OK && s->b.nonbinary)
            rc = tap_dyn_bits(s->ctx, s->d.B, s->n * s->R, s->rows, dyn_in, nullptr, s->b.nonbinary, stream);
        if (rc == TAP_OK)
            rc = tap_update_mask(s->ctx, s->d.B, s->n, s->R, nullptr, s->b.colsum[1], nullptr, s->b.current[1], s->b.mask[1], stream);
        if (rc != TAP_OK) { s->static_ = nullptr; return rc; }
        s->mask0 = s->b.mask[1];
        return TAP_OK;
    }
    if (!(flags & TAP_SB_INITIAL_MASK) || s->d.B == 0) return TAP_OK;
    // shadow + the masks DRL.forward starts from (model.py:297-307) in one launch that reads the tensor once; they
    // land in phase 1, which step 0 (writing phase 0) reads
    DeviceGuard g(s->ctx->device);
    const int rc = tap_mask_step_first(s->ctx, s->d.B, s->n, s->R, s->rows, 0, dyn_in, nullptr, 0, nullptr, nullptr,
                                       s->b.bits[1], nullptr, s->b.current[1], s->b.mask[1], s->b.nonbinary, stream);
    if (rc != TAP_OK) { s->static_ = nullptr; return rc; }
    s->bits0 = s->b.bits[1];
    s->mask0 = s->b.mask[1];
    return TAP_OK;
}

extern "C" int tap_stepper_begin_shadow(tap_stepper *s, const float *static_, const unsigned long long *bits, int flags)
{
    if (!s) return TAP_E_INVALID;
    if (s->copy_form) return tap_fail(s->ctx, TAP_E_UNSUPPORTED, "this window has no bit shadow: begin with tap_stepper_begin");
    if (s->d.B > 0 && (!static_ || !bits || bits == s->b.bits[0]))
        return tap_fail(s->ctx, TAP_E_INVALID, "stepper_begin_shadow needs static and a shadow that is not phase 0's buffer");
    if (s->d.B == 0) static_ = reinterpret_cast<const float *>(s);
    s->static_ = static_;
    s->dyn_in = nullptr;
    s->bits0 = bits;
    s->mask0 = nullptr;
    s->k = 0;
    s->keep = (flags & TAP_SB_CONTINUE) != 0;
    return TAP_OK;
}

extern "C" int tap_stepper_step(tap_stepper *s, const int64_t *ptr, void *stream)
{
    if (!s) return TAP_E_INVALID;
    if (!s->static_) return tap_fail(s->ctx, TAP_E_INVALID, "tap_stepper_begin has not been called");
    if (s->k >= s->steps) return tap_fail(s->ctx, TAP_E_STEPS, "the episode already took its %d steps", s->steps);
    if (s->d.B == 0) { s->k += 1; return TAP_OK; }
    const int k = s->k, w = k & 1, r = w ^ 1;
    const int flags = ((k == 0 && !s->keep) ? TAP_T_FRESH : 0) | (k == s->steps - 1 ? TAP_T_RATIO : 0);
    const StepAux aux = {s->b.decoder_static, s->b.tour, s->b.tour_stride > 0 ? s->b.tour_stride : s->steps, s->b.tour_col0 + k};
    DeviceGuard g(s->ctx->device);
    int rc;
    if (s->copy_form)
        rc = transition_copy_impl(s->ctx, &s->d, s->state, s->n, s->R, s->rows, s->update_rows, k == 0 ? s->dyn_in : s->b.dyn[r],
                                  s->static_, s->static_rows, ptr, k == 0 ? s->mask0 : s->b.mask[r], s->b.colsum[k == 0 ? 1 : r],
                                  s->b.dyn[w], s->b.colsum[w], s->b.current[w], s->b.mask[w], s->b.feature, s->b.ratio, flags,
                                  stream, &aux);
    else if (k == 0 && !s->bits0)
        rc = transition_first_impl(s->ctx, &s->d, s->state, s->n, s->R, s->rows, s->update_rows, s->dyn_in, s->static_,
                                   s->static_rows, ptr, nullptr, s->b.bits[w], s->b.dyn[w], s->b.current[w], s->b.mask[w],
                                   s->b.feature, s->b.ratio, s->b.nonbinary, flags, stream, &aux);
    else
        rc = transition_bits_impl(s->ctx, &s->d, s->state, s->n, s->R, s->rows, s->update_rows,
                                  k == 0 ? s->bits0 : s->b.bits[r], s->static_, s->static_rows, ptr,
                                  k == 0 ? s->mask0 : s->b.mask[r], s->b.bits[w], s->b.dyn[w], s->b.current[w], s->b.mask[w],
                                  s->b.feature, s->b.ratio, flags, stream, &aux, (k > 0 && s->inplace) ? 1 : 0);
    if (rc == TAP_OK) s->k = k + 1;
    return rc;
}

extern "C" int tap_stepper_steps_done(const tap_stepper *s) { return s ? s->k : TAP_E_INVALID; }
