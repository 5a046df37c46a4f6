// transition_wide.hip -- the fused step of transition.hip (update_dynamic + update_mask + gather + add_new_block
// in one launch, pack.py:276-376, model.py:404-465) for MACS / MUL 2D on containers of 17 .. 64 columns
// (tap_macs_wide.h: 32 / 64 lanes per container, the height-map in LDS).  Same workgroup shape as the other fused
// kernels: 8 (G = 32) or 4 (G = 64) envs, their placement waves at raised priority beside 4 stream waves.  A
// separate translation unit so that it compiles beside transition.hip.  gfx950 only.
#include "tap_common.h"
#include "tap_macs_wide.h"
#include "tap_transition.h"
#include "tap_waves.h"

template <int G, int NC, int MODE>
__global__ void __launch_bounds__((TransGeom<G, 4>::THREADS)) k_transition_macs_wide(TransArgs a)
{
    using Geo = TransGeom<G, 4>;
    constexpr int EPB = Geo::EPB, SPW = Geo::SPW, ENV_WAVES = Geo::ENV_WAVES;
    extern __shared__ float trans_lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int env_base = blockIdx.x * EPB;
    if (wave >= ENV_WAVES) {
        trans_stream_wave<SPW, NC, MODE>(a.m, env_base + (wave - ENV_WAVES) * SPW, lane,
                                         trans_lds + (size_t)(wave - ENV_WAVES) * SPW * 3 * a.m.nR);
        return;
    }
    __builtin_amdgcn_s_setprio(2);
    int *macs_base = reinterpret_cast<int *>(trans_lds + (size_t)EPB * 3 * a.m.nR);
    tap_macs_wide_wave<G>(a.s, a.flags, a.ratio_out, env_base + tid / G, tid % G, lane,
                          macs_base + (tid / G) * macs_wide_group_words(G, a.s.d.H, a.s.d.n_max, a.s.d.W));
}

template <int G> static size_t transition_macs_wide_lds(const tap_env_desc &d, int nR)
{
    constexpr int EPB = TransGeom<G, 4>::EPB;
    // the stream waves' tiles (3 * nR floats per env), then the placement groups' slices, 8-byte aligned
    return (size_t)EPB * 3 * nR * sizeof(float) + (size_t)EPB * macs_wide_group_words(G, d.H, d.n_max, d.W) * sizeof(int);
}

// does the step fit one workgroup's LDS?  (transition.hip falls back to the two launches when it does not)
bool tap_transition_macs_wide_fits(const tap_env_desc *d, int nR)
{
    const size_t lds = d->W > 32 ? transition_macs_wide_lds<64>(*d, nR) : transition_macs_wide_lds<32>(*d, nR);
    return lds <= 64 * 1024;
}

template <int G> static int launch_transition_macs_wide(tap_ctx *ctx, const TransArgs &a, hipStream_t st)
{
    constexpr int EPB = TransGeom<G, 4>::EPB, THREADS = TransGeom<G, 4>::THREADS;
    const int grid = (a.s.d.B + EPB - 1) / EPB;
    if (grid == 0) return TAP_OK;
    const size_t lds = transition_macs_wide_lds<G>(a.s.d, a.m.nR);
    if (lds > 64 * 1024) return tap_fail(ctx, TAP_E_UNSUPPORTED, "transition(MACS, wide): %zu bytes of LDS needed", lds);
    const int mode = a.m.bits_in ? 1 : mask_builds_bits(a.m) ? 2 : 0;
#define TAP_LAUNCH_T(NC_, M_, LDS_) hipLaunchKernelGGL((k_transition_macs_wide<G, NC_, M_>), dim3(grid), dim3(THREADS), LDS_, st, a)
#define TAP_LAUNCH_M(NC_, LDS_) do { if (mode == 1) TAP_LAUNCH_T(NC_, 1, LDS_); else if (mode == 2) TAP_LAUNCH_T(NC_, 2, LDS_); else TAP_LAUNCH_T(NC_, 0, LDS_); } while (0)
    switch (mask_fast_path_cols(a.m)) {
    case 1: TAP_LAUNCH_M(1, lds); break;
    case 2: TAP_LAUNCH_M(2, lds); break;
    case 4: TAP_LAUNCH_M(4, lds); break;
    default: TAP_LAUNCH_T(0, 0, lds); break;
    }
#undef TAP_LAUNCH_M
#undef TAP_LAUNCH_T
    TAP_LAUNCH_CHECK(ctx, "k_transition_macs_wide");
    return TAP_OK;
}

int tap_transition_macs_wide(tap_ctx *ctx, const TransArgs &a, hipStream_t st)
{
    return a.s.d.W > 32 ? launch_transition_macs_wide<64>(ctx, a, st) : launch_transition_macs_wide<32>(ctx, a, st);
}
