// macs.hip -- stand-alone MACS / MUL step (tools.Container.add_new_block with packing_strategy
// 'MACS' / 'MUL'); the placement itself is tap_macs.h (2D) / tap_macs3.h (3D).  gfx950 only.
#include <cstdlib>

#include "tap_common.h"
#include "tap_macs.h"
#include "tap_macs_wide.h"
#include "tap_macs3.h"

template <int G>
__global__ void __launch_bounds__(TAP_BLOCK) k_macs2d_step(StepArgs a)
{
    extern __shared__ int lds[];
    const int tid = threadIdx.x, cell = tid % G;
    const int env = blockIdx.x * ((int)blockDim.x / G) + tid / G;
    const int B = a.d.B, W = a.d.W, H = a.d.H;
    const bool ev = env < B, incell = cell < W;
    const int gl0 = (tid & 63) - cell;
    const MacsLds L = macs_lds(lds + (tid / G) * macs_group_words(G, H, a.d.n_max, W), G, H, macs_ems_cap(W, a.d.n_max));

    int hm = (ev && incell) ? a.v.hm[(size_t)env * W + cell] : 0;
    const int cv = (ev && cell < 4) ? a.v.cnt[(size_t)env * 4 + cell] : 0;
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
    const Placement pl = tap_macs_place<G>(cfg, L, cell, gl0, hm, cnt, err, bx, bz, do_step);
    err = group_or<G>(err);

    tap_wave_lds_sync();
    L.hm[cell] = hm;
    tap_wave_lds_sync();
    if (ev) {
        if (incell) a.v.hm[(size_t)env * W + cell] = hm;
        if (a.feature_out)
            tap_write_feature<2, G>(a.d.feature, W, 1, L.hm, cell, hm, a.feature_out + (size_t)env * a.flen);
        if (cell == 0) {
            if (do_step) {
                reinterpret_cast<int4 *>(a.v.cnt)[env] = make_int4(cnt.valid, cnt.empty, cnt.nstable, cnt.count);
                a.v.pos[(size_t)(step * 2) * B + env] = pl.x;
                a.v.pos[(size_t)(step * 2 + 1) * B + env] = pl.z;
                a.v.stable[(size_t)step * B + env] = (uint8_t)pl.stab;
                a.v.blk[(size_t)(step * 2) * B + env] = bx;   // history the later steps read
                a.v.blk[(size_t)(step * 2 + 1) * B + env] = bz; // (tools.py:2531-2533), failures too
            }
            if (err) a.v.err[env] |= err;
        }
    } else if (a.d.feature == TAP_FEAT_ZERO) {
        (void)group_min<G>(INT_MAX);
    }
}

template <int G> static int launch_macs(tap_ctx *ctx, const StepArgs &a, hipStream_t st)
{
    const tap_env_desc &d = a.d;
    const size_t per_env = (size_t)macs_group_words(G, d.H, d.n_max, d.W) * sizeof(int);
    const int threads = tap_lds_threads(per_env, G, tap_lds_limit(ctx));   // containers per workgroup by the device's LDS
    if (threads == 0)
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "MACS: H=%d blocks_num=%d need %zu bytes of LDS per container", d.H, d.n_max, per_env);
    const int epb = threads / G, grid = (d.B + epb - 1) / epb;
    if (grid == 0) return TAP_OK;
    const size_t lds = epb * per_env;
    TAP_HIP_CHECK(ctx, tap_allow_lds(k_macs2d_step<G>, lds));
    hipLaunchKernelGGL(k_macs2d_step<G>, dim3(grid), dim3(threads), lds, st, a);
    TAP_LAUNCH_CHECK(ctx, "k_macs2d_step");
    return TAP_OK;
}

// ---- 2D, 17 .. 64 columns (tap_macs_wide.h): height-map in LDS, 64-bit column masks ---------------------
template <int G>
__global__ void __launch_bounds__(TAP_BLOCK) k_macs2d_wide_step(StepArgs a)
{
    extern __shared__ int lds[];
    const int tid = threadIdx.x;
    tap_macs_wide_wave<G>(a, 0, nullptr, blockIdx.x * ((int)blockDim.x / G) + tid / G, tid % G, tid & 63,
                          lds + (tid / G) * macs_wide_group_words(G, a.d.H, a.d.n_max, a.d.W));
}

template <int G> static int launch_macs_wide(tap_ctx *ctx, const StepArgs &a, hipStream_t st)
{
    const tap_env_desc &d = a.d;
    const size_t per_env = (size_t)macs_wide_group_words(G, d.H, d.n_max, d.W) * sizeof(int);
    const int threads = tap_lds_threads(per_env, G, tap_lds_limit(ctx));
    if (threads == 0)
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "MACS: W=%d H=%d blocks_num=%d need %zu bytes of LDS per container", d.W, d.H, d.n_max, per_env);
    const int epb = threads / G, grid = (d.B + epb - 1) / epb;
    if (grid == 0) return TAP_OK;
    const size_t lds = epb * per_env;
    TAP_HIP_CHECK(ctx, tap_allow_lds(k_macs2d_wide_step<G>, lds));
    hipLaunchKernelGGL(k_macs2d_wide_step<G>, dim3(grid), dim3(threads), lds, st, a);
    TAP_LAUNCH_CHECK(ctx, "k_macs2d_wide_step");
    return TAP_OK;
}

// ---- 3D ------------------------------------------------------------------------------------------
template <int G, int WL = 0>                            // WL: compile-time sides (tap_macs3_place), the reference's 5 x 5
__global__ void __launch_bounds__(TAP_BLOCK) k_macs3d_step(StepArgs a)
{
    extern __shared__ int lds[];
    const int tid = threadIdx.x;
    tap_macs3_wave<G, WL>(a, 0, nullptr, blockIdx.x * ((int)blockDim.x / G) + tid / G, tid % G, tid & 63,
                      lds + (tid / G) * macs3_group_words(G, a.d.n_max, a.d.H));
}

template <int G> static int launch_macs3(tap_ctx *ctx, const StepArgs &a, hipStream_t st)
{
    const tap_env_desc &d = a.d;
    const size_t per_env = (size_t)macs3_group_words(G, d.n_max, d.H) * sizeof(int);
    const int threads = tap_lds_threads(per_env, G, tap_lds_limit(ctx));
    if (threads == 0)
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "MACS 3D: H=%d blocks_num=%d need %zu bytes of LDS per container", d.H, d.n_max, per_env);
    const int epb = threads / G, grid = (d.B + epb - 1) / epb;
    if (grid == 0) return TAP_OK;
    const size_t lds = epb * per_env;
    if constexpr (G == 32) {
        if (d.W == 5 && d.L == 5) {
            TAP_HIP_CHECK(ctx, tap_allow_lds(k_macs3d_step<G, 5>, lds));
            hipLaunchKernelGGL((k_macs3d_step<G, 5>), dim3(grid), dim3(threads), lds, st, a);
            TAP_LAUNCH_CHECK(ctx, "k_macs3d_step");
            return TAP_OK;
        }
    }
    TAP_HIP_CHECK(ctx, tap_allow_lds(k_macs3d_step<G>, lds));
    hipLaunchKernelGGL(k_macs3d_step<G>, dim3(grid), dim3(threads), lds, st, a);
    TAP_LAUNCH_CHECK(ctx, "k_macs3d_step");
    return TAP_OK;
}

int tap_macs_validate(tap_ctx *ctx, const tap_env_desc &d)
{
    if (d.D == 3) {
        if (d.W > 64 || d.L > 64 || d.H > MACS3_MAX_H)
            return tap_fail(ctx, TAP_E_UNSUPPORTED, "MACS 3D supports W, L <= 64 and H <= %d", MACS3_MAX_H);
        return TAP_OK;
    }
    if (d.W > 4096 || d.H > MACS_MAX_H)
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "MACS supports W <= 4096 and H <= %d", MACS_MAX_H);
    return TAP_OK;
}

// from which width MACS 2D runs one wavefront per container (TAP_MACS2D_WAVE_FROM=W moves the hand-over for A/B runs)
int tap_macs2d_wave_from() { static const int from = [] { const char *e = getenv("TAP_MACS2D_WAVE_FROM"); return e ? atoi(e) : 17; }(); return from; }

int tap_macs2d_step(tap_ctx *ctx, const StepArgs &a, hipStream_t st)
{
    int rc = tap_macs_validate(ctx, a.d);
    if (rc) return rc;
    if (a.d.D == 3) {
        if (tap_is_big_macs3(&a.d)) return tap_macs3_big_step(ctx, a, st);  // one wavefront per container (macs3_big.hip)
        switch (tap_group_size(&a.d)) {
        case 8: return launch_macs3<8>(ctx, a, st);
        case 16: return launch_macs3<16>(ctx, a, st);
        case 32: return launch_macs3<32>(ctx, a, st);
        default: return launch_macs3<64>(ctx, a, st);
        }
    }
    if (a.d.W > 64) return tap_macs_big_step(ctx, a, st);              // one wavefront per container (macs_big.hip)
    {   // above 16 columns the wave-per-container kernel of macs_big.hip beats the 32- and 64-lane forms of
        // tap_macs_wide.h, which stay as its fallback when a tile does not fit the LDS (eager steps, n = 10: at B = 4096
        // W = 17 35 against 48 us, W = 32 35 against 120; at B = 65 536 W = 17 282 against 304, W = 32 281 against 1 696;
        // W = 16 35.5 against 33.4 on the lane kernel); TAP_MACS2D_WAVE_FROM=W moves the hand-over for A/B runs
        const int from = tap_macs2d_wave_from();
        if (a.d.W >= from) {
            const int rc_w = tap_macs_wave_step(ctx, a, st);
            if (rc_w != TAP_E_UNSUPPORTED) return rc_w;                  // launched, or a real error; only "the tile does not fit" falls back
        }
    }
    if (a.d.W > 32) return launch_macs_wide<64>(ctx, a, st);
    if (a.d.W > 16) return launch_macs_wide<32>(ctx, a, st);
    return a.d.W <= 8 ? launch_macs<8>(ctx, a, st) : launch_macs<16>(ctx, a, st);
}

#ifdef TAP_PROF
extern "C" int tap_prof_read_macs2(unsigned int *out)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(tap_prof_m2), sizeof(unsigned int) * 8192 * 8);
    return 0;
}
extern "C" int tap_prof_read_macs3(unsigned int *out)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(tap_prof_m3), sizeof(unsigned int) * 8192 * 16);
    return 0;
}
#endif
