// transition.hip -- one whole lock-step of DRL.forward's environment work in a single launch:
// update_dynamic + update_mask (pack.py:276-376, model.py:376-386) and the gather + add_new_block
// of model.py:404-465, optionally starting from a fresh container (Container.__init__) and
// optionally emitting calc_ratio (model.py:499-510).  gfx950 only.
//
// Why fuse: at the BASELINE batch sizes the placement moves ~0.9 MB per launch and is pure latency
// (two dependent loads, fp64 divides), while the precedence update streams ~40 MB and is HBM-bound.
// In one launch the placement's latency hides under the stream and a kernel boundary per step
// disappears.  A 256-thread workgroup owns EPB = 8 consecutive envs (4 when an env needs 64 lanes):
// every wave streams EPB/4 of their dynamic slabs with 16-byte accesses, and the first EPB*G/64
// waves also carry the lane-per-cell placement groups.  Env lanes issue their loads first, then the
// wave streams, then the placement computes on data that has long arrived.  No s_barrier anywhere:
// lane groups never span a wave, so LDS hand-offs only need compiler ordering.
#include "tap_common.h"
#include "tap_masks.h"
#include "tap_place.h"

struct TransArgs {
    StepArgs s;   // placement (always the gather form: s.static_, s.ptr)
    MaskArgs m;   // precedence update
    int flags;
    float *ratio_out;
};

template <int D, int G, int VEC>
__global__ void __launch_bounds__(TAP_BLOCK) k_transition(TransArgs a)
{
    constexpr int EPB = (G == 64) ? 4 : 8;      // envs per workgroup
    constexpr int SPW = EPB / 4;                // slabs per wave
    constexpr int ENV_WAVES = EPB * G / 64;     // waves that carry placement lanes
    __shared__ int s_old[TAP_BLOCK];
    __shared__ int s_new[TAP_BLOCK];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int env_base = blockIdx.x * EPB;
    const int B = a.s.d.B, W = a.s.d.W, L = a.s.d.L, cells = W * L;
    const bool fresh = a.flags & TAP_T_FRESH;

    // ---- placement lanes: issue the loads ------------------------------------------------------
    const int grp = tid / G, cell = tid % G;
    const int env = env_base + grp;
    const bool ev = (wave < ENV_WAVES) && env < B, incell = cell < cells;
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

    // ---- every wave: stream its slabs (pack.py:370-374) -------------------------------------------
    const size_t slab = (size_t)a.m.rows * a.m.nR;
    long real[SPW], pp[SPW];
#pragma unroll
    for (int k = 0; k < SPW; ++k) {
        const int senv = env_base + wave * SPW + k;
        real[k] = -1; pp[k] = 0;
        if (senv >= B) continue;
        pp[k] = (long)a.m.ptr[senv];
        real[k] = (long)a.m.static_[(size_t)senv * a.m.static_rows * a.m.nR + pp[k]]; // pack.py:339
        const ClearRanges cr = clear_ranges(a.m, real[k]);
        const float *src = a.m.dyn_in + (size_t)senv * slab;
        float *dst = a.m.dyn_out + (size_t)senv * slab;
        if (VEC == 4) {
            const int nchunk = (int)(slab / 4);
            const float4 *s4 = reinterpret_cast<const float4 *>(src);
            float4 *d4 = reinterpret_cast<float4 *>(dst);
#pragma unroll 8
            for (int q = lane; q < nchunk; q += 64) {
                float4 v = s4[q];
                if (in_cleared(cr, (long)q * 4)) v = make_float4(0.f, 0.f, 0.f, 0.f);
                d4[q] = v;
            }
        } else {
            for (long f = lane; f < (long)slab; f += 64) {
                float v = src[f];
                if (in_cleared(cr, f)) v = 0.f;
                dst[f] = v;
            }
        }
    }

    // ---- placement (tools.py:3663-3744) -----------------------------------------------------------
    if (wave < ENV_WAVES) { // wave-uniform
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

    // ---- column sums + masks of the slabs this wave streamed (pack.py:318-329) ---------------------
#pragma unroll
    for (int k = 0; k < SPW; ++k) {
        const int senv = env_base + wave * SPW + k;
        if (senv < B) mask_env(a.m, senv, lane, real[k], pp[k]);
    }
}

template <int D, int G> static int launch_transition(tap_ctx *ctx, const TransArgs &a, hipStream_t st)
{
    constexpr int EPB = (G == 64) ? 4 : 8;
    const int grid = (a.s.d.B + EPB - 1) / EPB;
    if (grid == 0) return TAP_OK;
    const bool vec = (a.m.nR % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(a.m.dyn_in) | reinterpret_cast<uintptr_t>(a.m.dyn_out)) % 16 == 0);
    if (vec) hipLaunchKernelGGL((k_transition<D, G, 4>), dim3(grid), dim3(TAP_BLOCK), 0, st, a);
    else hipLaunchKernelGGL((k_transition<D, G, 1>), dim3(grid), dim3(TAP_BLOCK), 0, st, a);
    TAP_LAUNCH_CHECK(ctx, "k_transition");
    return TAP_OK;
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
