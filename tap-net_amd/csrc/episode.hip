// episode.hip -- whole episodes in ONE launch: every container packs its whole block list from empty, state in
// registers / LDS, nothing but the inputs and the per-episode results touches HBM.
//   tools.calc_positions_lb_greedy (tools.py:2393-2449)  -> k_episode<D,G>          (tap_place.h)
//   tools.calc_positions_mcs       (tools.py:3213-3315)  -> k_episode_macs2<G,WIDE> (tap_macs.h / tap_macs_wide.h)
//                                                           k_episode_macs3<G>      (tap_macs3.h)
// Callers in the reference: pack.reward (pack.py:378-473), pack.render (pack.py:743-792), generate.generate_blocks
// (generate.py:908), generate_blocks_with_GT (generate.py:112).  gfx950 only.
#include "tap_common.h"
#include "tap_place.h"
#include "tap_macs.h"
#include "tap_macs_wide.h"
#include "tap_macs3.h"
#include "tap_episode.h"

// The block lists are read PF steps at a time by the lanes of the container's group (two dependent loads -- tour
// entry, then the column of `static` -- for PF steps at once) into an LDS row, instead of two dependent round trips
// in front of every placement: at ten blocks an episode's 20 serial memory latencies become 2.
constexpr int EP_PF = 32;
template <int D, int G>
__device__ __forceinline__ void episode_prefetch(const EpisodeArgs &a, int env, bool ev, int cell, int t0, int4 *row)
{
    for (int j = cell; j < EP_PF && t0 + j < a.n; j += G) {
        int dims[3], e = 0;
        const bool in = episode_block<D>(a, env, t0 + j, ev, dims, e);
        row[j] = make_int4(dims[0], dims[1], dims[2], (in ? 1 : 0) | (e << 1));   // w: in-list flag, error bits above
    }
}

// ---- LB_GREEDY ---------------------------------------------------------------------------------------------
// SOFT: the caller has checked that TAP_F_HARD is not set; the hard-mode walk is not compiled in (3D: 134 -> ~60
// VGPRs).  Lane groups never span a wavefront, so the per-step hand-offs are wave-level LDS syncs, not s_barrier.
template <int D, int G, bool SOFT>
__global__ void __launch_bounds__(TAP_BLOCK) k_episode(EpisodeArgs a)
{
    __shared__ int s[TAP_BLOCK];
    __shared__ int4 pf[TAP_BLOCK / G][EP_PF];
    const int tid = threadIdx.x, grp = tid / G, cell = tid % G;
    const int env = blockIdx.x * ((int)blockDim.x / G) + grp;
    const int W = a.d.W, L = a.d.L, n = a.n;
    const bool ev = env < a.B, incell = cell < W * L;
    const PlaceCfg cfg = {W, L, a.d.H, a.d.flags, a.lut};
    int hm = 0, err = 0;
    Counters cnt = {0, 0, 0, 0};
    for (int t0 = 0; t0 < n; t0 += EP_PF) {
        tap_wave_lds_sync();                                  // the previous rows have been consumed
        episode_prefetch<D, G>(a, env, ev, cell, t0, pf[grp]);
        tap_wave_lds_sync();
        for (int j = 0; j < EP_PF && t0 + j < n; ++j) {
            const int t = t0 + j;
            const int4 b = pf[grp][j];
            const int bx = b.x, by = D == 3 ? b.y : 1, bz = D == 3 ? b.z : b.y;
            err |= b.w >> 1;
            s[tid] = hm;
            tap_wave_lds_sync();
            const Placement pl = tap_place<D, G, !SOFT>(cfg, s + grp * G, cell, hm, cnt, err, bx, by, bz, (b.w & 1) != 0);
            tap_wave_lds_sync();
            if (ev && cell == 0) {
                if (a.pos_out) {
                    int32_t *pp = a.pos_out + ((size_t)env * n + t) * D;
                    pp[0] = pl.x;
                    if (D == 3) { pp[1] = pl.y; pp[2] = pl.z; } else pp[1] = pl.z;
                }
                if (a.stable_out) a.stable_out[(size_t)env * n + t] = (uint8_t)pl.stab;
            }
        }
    }
    const int gmax = group_max<G>(incell ? hm : 0);
    err = group_or<G>(err);
    if (ev && cell == 0) episode_finish(a, env, cnt, gmax, err);
}

template <int D, int G> static int launch_episode(tap_ctx *ctx, const EpisodeArgs &a, hipStream_t st)
{
    const int epb = TAP_BLOCK / G, grid = (a.B + epb - 1) / epb;
    if (grid == 0) return TAP_OK;
    if (a.d.flags & TAP_F_HARD) hipLaunchKernelGGL((k_episode<D, G, false>), dim3(grid), dim3(TAP_BLOCK), 0, st, a);
    else hipLaunchKernelGGL((k_episode<D, G, true>), dim3(grid), dim3(TAP_BLOCK), 0, st, a);
    TAP_LAUNCH_CHECK(ctx, "k_episode");
    return TAP_OK;
}

// ---- MACS / MUL 2D: the step of macs.hip with the state kept in LDS across the steps -------------------------
template <int G, bool WIDE>
__global__ void __launch_bounds__(TAP_BLOCK) k_episode_macs2(EpisodeArgs a)
{
    extern __shared__ int lds[];
    const int tid = threadIdx.x, cell = tid % G;
    const int env = blockIdx.x * ((int)blockDim.x / G) + tid / G;
    const int W = a.d.W, H = a.d.H, n = a.n;
    const bool ev = env < a.B, incell = cell < W;
    const int gl0 = (tid & 63) - cell;
    const int cap = macs_ems_cap(W, n);
    const int words = WIDE ? macs_wide_group_words(G, H, n, W) : macs_group_words(G, H, n, W);
    int *base = lds + (tid / G) * words;
    const MacsLds Ln = macs_lds(base, G, H, cap);
    const MacsWideLds Lw = macs_wide_lds(base, G, H, cap);
    int *hist = WIDE ? Lw.hist : Ln.hist;
    const PlaceCfg cfg = {W, 1, H, a.d.flags, nullptr};
    int hm = 0, err = 0;
    Counters cnt = {0, 0, 0, 0};
    __shared__ int4 pf[TAP_BLOCK / G][EP_PF];
    int4 *const row = pf[tid / G];
    for (int t = 0; t < n; ++t) {
        if ((t & (EP_PF - 1)) == 0) {                         // t is uniform: every lane takes this branch together
            tap_wave_lds_sync();
            episode_prefetch<2, G>(a, env, ev, cell, t, row);
            tap_wave_lds_sync();
        }
        const int4 b = row[t & (EP_PF - 1)];
        const bool in = (b.w & 1) != 0;
        err |= b.w >> 1;
        const int bx = b.x, bz = b.y;
        bool do_step = in;
        if (in && (bx < 1 || bz < 1)) { err |= 4; do_step = false; }
        base[cell] = hm;                                      // L.hm
        if constexpr (WIDE) { for (int i = cell; i < H; i += G) Lw.taken[i] = 0; }
        else { for (int i = cell; i < H; i += G) Ln.taken[i] = 0; }
        tap_wave_lds_sync();
        const int step = cnt.count;
        Placement pl = {0, 0, 0, 0, 0};
        if constexpr (WIDE) pl = tap_macs_place_wide<G>(cfg, Lw, cell, gl0, hm, cnt, err, bx, bz, do_step);
        else pl = tap_macs_place<G>(cfg, Ln, cell, gl0, hm, cnt, err, bx, bz, do_step);
        tap_wave_lds_sync();
        if (do_step && cell == 0) {                           // history the later steps read (tools.py:2531-2533),
            hist[step * 4] = pl.x; hist[step * 4 + 1] = pl.z; // failures too, at (0, 0)
            hist[step * 4 + 2] = bx; hist[step * 4 + 3] = bz;
        }
        if (ev && cell == 0) {
            if (a.pos_out) { a.pos_out[((size_t)env * n + t) * 2] = pl.x; a.pos_out[((size_t)env * n + t) * 2 + 1] = pl.z; }
            if (a.stable_out) a.stable_out[(size_t)env * n + t] = (uint8_t)pl.stab;
        }
        tap_wave_lds_sync();
    }
    const int gmax = group_max<G>(incell ? hm : 0);
    err = group_or<G>(err);
    if (ev && cell == 0) episode_finish(a, env, cnt, gmax, err);
}

template <int G, bool WIDE> static int launch_episode_macs2(tap_ctx *ctx, const EpisodeArgs &a, hipStream_t st)
{
    const tap_env_desc &d = a.d;
    const size_t per_env = (size_t)(WIDE ? macs_wide_group_words(G, d.H, a.n, d.W) : macs_group_words(G, d.H, a.n, d.W)) * sizeof(int);
    // the block-list rows are static LDS
    const int threads = tap_lds_threads(per_env, G, tap_lds_limit(ctx), (TAP_BLOCK / G) * EP_PF * sizeof(int4));
    if (threads == 0)
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "MACS episode: W=%d H=%d n=%d need %zu bytes of LDS per container", d.W, d.H, a.n, per_env);
    const int epb = threads / G, grid = (a.B + epb - 1) / epb;
    if (grid == 0) return TAP_OK;
    const size_t lds = epb * per_env;
    TAP_HIP_CHECK(ctx, tap_allow_lds(k_episode_macs2<G, WIDE>, lds));
    hipLaunchKernelGGL((k_episode_macs2<G, WIDE>), dim3(grid), dim3(threads), lds, st, a);
    TAP_LAUNCH_CHECK(ctx, "k_episode_macs2");
    return TAP_OK;
}

// ---- MACS / MUL 3D ---------------------------------------------------------------------------------------------
template <int G, int WL = 0>                            // WL: compile-time sides (tap_macs3_place), the reference's 5 x 5
__global__ void __launch_bounds__(TAP_BLOCK) k_episode_macs3(EpisodeArgs a)
{
    extern __shared__ int lds[];
    const int tid = threadIdx.x, cell = tid % G;
    const int env = blockIdx.x * ((int)blockDim.x / G) + tid / G;
    const int W = WL ? WL : a.d.W, Ld = WL ? WL : a.d.L, cells = W * Ld, n = a.n;
    const bool ev = env < a.B, incell = cell < cells;
    const int gl0 = (tid & 63) - cell;
    const int HW = macs3_hw(a.d.H);
    const Macs3Lds S = macs3_lds(lds + (tid / G) * macs3_group_words(G, n, a.d.H), G, a.d.H);
    const PlaceCfg cfg = {W, Ld, a.d.H, a.d.flags, a.lut};
    for (int k = cell; k < G * HW; k += G) S.occ[k] = 0ull;   // level_free_space of an empty container
    for (int k = cell; k < 256; k += G) S.lrun[k] = (unsigned char)m3_longest_run((unsigned)k);
    int hm = 0, err = 0;
    Counters cnt = {0, 0, 0, 0};
    __shared__ int4 pf[TAP_BLOCK / G][EP_PF];
    int4 *const row = pf[tid / G];
    for (int t = 0; t < n; ++t) {
        if ((t & (EP_PF - 1)) == 0) {
            tap_wave_lds_sync();
            episode_prefetch<3, G>(a, env, ev, cell, t, row);
            tap_wave_lds_sync();
        }
        const int4 b = row[t & (EP_PF - 1)];
        const bool in = (b.w & 1) != 0;
        err |= b.w >> 1;
        const int bx = b.x, by = b.y, bz = b.z;
        bool do_step = in;
        // sides larger than the container: see tap_macs3_wave
        if (in && (bx < 1 || by < 1 || bz < 1 || bx > W || by > Ld)) { err |= 4; do_step = false; }
        S.hm[cell] = hm;
        tap_wave_lds_sync();
        const int step = cnt.count;
        const Placement pl = tap_macs3_place<G, WL>(cfg, S, cell, gl0, hm, cnt, err, bx, by, bz, do_step);
        tap_wave_lds_sync();
        if (do_step && cell == 0) {                           // tools.py:2843-2846: failures too
            S.hist[step * MACS3_HIST] = (pl.x & 15) | ((pl.y & 15) << 4) | ((bx & 15) << 8) | ((by & 15) << 12) | ((pl.placed & 1) << 16);
            S.hist[step * MACS3_HIST + 1] = (pl.z & 0xffff) | (bz << 16);
        }
        if (ev && cell == 0) {
            if (a.pos_out) {
                int32_t *pp = a.pos_out + ((size_t)env * n + t) * 3;
                pp[0] = pl.x; pp[1] = pl.y; pp[2] = pl.z;
            }
            if (a.stable_out) a.stable_out[(size_t)env * n + t] = (uint8_t)pl.stab;
        }
        tap_wave_lds_sync();
    }
    const int gmax = group_max<G>(incell ? hm : 0);
    err = group_or<G>(err);
    if (ev && cell == 0) episode_finish(a, env, cnt, gmax, err);
}

template <int G> static int launch_episode_macs3(tap_ctx *ctx, const EpisodeArgs &a, hipStream_t st)
{
    const tap_env_desc &d = a.d;
    const size_t per_env = (size_t)macs3_group_words(G, a.n, d.H) * sizeof(int);
    const int threads = tap_lds_threads(per_env, G, tap_lds_limit(ctx), (TAP_BLOCK / G) * EP_PF * sizeof(int4));
    if (threads == 0)
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "MACS 3D episode: H=%d n=%d need %zu bytes of LDS per container", d.H, a.n, per_env);
    const int epb = threads / G, grid = (a.B + epb - 1) / epb;
    if (grid == 0) return TAP_OK;
    const size_t lds = epb * per_env;
    if constexpr (G == 32) {
        if (d.W == 5 && d.L == 5) {
            TAP_HIP_CHECK(ctx, tap_allow_lds(k_episode_macs3<G, 5>, lds));
            hipLaunchKernelGGL((k_episode_macs3<G, 5>), dim3(grid), dim3(threads), lds, st, a);
            TAP_LAUNCH_CHECK(ctx, "k_episode_macs3");
            return TAP_OK;
        }
    }
    TAP_HIP_CHECK(ctx, tap_allow_lds(k_episode_macs3<G>, lds));
    hipLaunchKernelGGL(k_episode_macs3<G>, dim3(grid), dim3(threads), lds, st, a);
    TAP_LAUNCH_CHECK(ctx, "k_episode_macs3");
    return TAP_OK;
}

int tap_macs_validate(tap_ctx *ctx, const tap_env_desc &d); // macs.hip

int tap_big_episode(tap_ctx *ctx, const EpisodeArgs &a, hipStream_t st);   // big.hip: LB_GREEDY above 64 cells

static int episode_dispatch(tap_ctx *ctx, const tap_env_desc *d, const EpisodeArgs &a, hipStream_t st)
{
    if (tap_is_big(d)) return tap_big_episode(ctx, a, st);
    if (d->strategy == TAP_MACS) {
        int rc = tap_macs_validate(ctx, *d);
        if (rc) return rc;
        if (d->D == 3) {
            switch (tap_group_size(d)) {
            case 8: return launch_episode_macs3<8>(ctx, a, st);
            case 16: return launch_episode_macs3<16>(ctx, a, st);
            case 32: return launch_episode_macs3<32>(ctx, a, st);
            default: return launch_episode_macs3<64>(ctx, a, st);
            }
        }
        if (d->W > 32) return launch_episode_macs2<64, true>(ctx, a, st);
        if (d->W > 16) return launch_episode_macs2<32, true>(ctx, a, st);
        return d->W <= 8 ? launch_episode_macs2<8, false>(ctx, a, st) : launch_episode_macs2<16, false>(ctx, a, st);
    }
    TAP_DISPATCH_DG(launch_episode, d, ctx, a, st);
}

static int episode_validate(tap_ctx *ctx, const tap_env_desc *d, const char *what)
{
    int rc = tap_desc_validate(ctx, d);
    if (rc) return rc;
    if (d->strategy == TAP_LB)
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "%s: the legacy 'LB' strategy has no whole-episode form in use (tools.calc_positions_greedy is commented out at pack.py:741)", what);
    if (tap_is_big_macs(d) || tap_is_big_macs3(d))
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "%s: MACS / MUL containers above 64 cells are stepped with tap_env_step_gather", what);
    return TAP_OK;
}

extern "C" int tap_episode_reward(tap_ctx *ctx, const tap_env_desc *d, int B, int n,
                                  const float *static_, int static_rows, int nR,
                                  const int64_t *tour, float *reward_out, int32_t *positions_out,
                                  uint8_t *stable_out, void *stream)
{
    int rc = episode_validate(ctx, d, "episode reward");
    if (rc) return rc;
    if (B == 0) return TAP_OK; // an empty batch has no buffers to check (d->B is ignored here)
    if (d->strategy != TAP_LB_GREEDY)
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "episode reward: the reference only defines LB_GREEDY here (pack.py:431 names a missing function); tap_episode_scores packs with MACS / MUL");
    if (!static_ || !tour || !reward_out || B < 0 || n < 1 || static_rows < 1 + d->D || nR < 1)
        return tap_fail(ctx, TAP_E_INVALID, "bad episode arguments");
    EpisodeArgs a = {};
    a.d = *d; a.d.ratio_mode = TAP_R_CPS;                      // tools.py:2442-2445: every type is C + P + S
    a.B = B; a.n = n; a.static_ = static_; a.static_rows = static_rows; a.nR = nR; a.tour = tour; a.target_sel = -1;
    a.lut = ctx ? ctx->stab_lut : nullptr; a.reward_out = reward_out; a.pos_out = positions_out; a.stable_out = stable_out;
    return episode_dispatch(ctx, d, a, (hipStream_t)stream);
}

extern "C" int tap_episode_scores(tap_ctx *ctx, const tap_env_desc *d, int B, int n, const float *static_,
                                  int static_rows, int nR, const int64_t *tour, int target_sel,
                                  double *ratio64_out, int64_t *scores_out, int32_t *positions_out,
                                  uint8_t *stable_out, int32_t *err_out, void *stream)
{
    int rc = episode_validate(ctx, d, "episode scores");
    if (rc) return rc;
    if (B == 0) return TAP_OK;
    if (!static_ || !tour || B < 0 || n < 1 || static_rows < 1 + d->D || nR < 1 || target_sel < -1 || target_sel > 1 ||
        (target_sel >= 0 && static_rows < 2 + d->D))
        return tap_fail(ctx, TAP_E_INVALID, "bad episode arguments");
    EpisodeArgs a = {};
    a.d = *d;
    if (d->strategy == TAP_LB_GREEDY || d->ratio_mode == TAP_R_CP_HALF) a.d.ratio_mode = TAP_R_CPS; // tools.py:2442-2445
    a.B = B; a.n = n; a.static_ = static_; a.static_rows = static_rows; a.nR = nR; a.tour = tour; a.target_sel = target_sel;
    a.lut = ctx ? ctx->stab_lut : nullptr; a.pos_out = positions_out; a.stable_out = stable_out;
    a.score64_out = ratio64_out; a.scores_out = scores_out; a.err_out = err_out;
    return episode_dispatch(ctx, d, a, (hipStream_t)stream);
}

extern "C" int tap_pack_blocks(tap_ctx *ctx, const tap_env_desc *d, int B, int n, const int32_t *blocks,
                               float *reward_out, int32_t *positions_out, uint8_t *stable_out,
                               double *score64_out, void *stream)
{
    int rc = episode_validate(ctx, d, "pack_blocks");
    if (rc) return rc;
    if (B == 0) return TAP_OK; // an empty batch has no buffers to check (d->B is ignored here)
    if (!blocks || B < 0 || n < 1) return tap_fail(ctx, TAP_E_INVALID, "bad pack_blocks arguments");
    EpisodeArgs a = {};
    a.d = *d;
    if (d->strategy == TAP_LB_GREEDY || d->ratio_mode == TAP_R_CP_HALF) a.d.ratio_mode = TAP_R_CPS;
    a.B = B; a.n = n; a.blocks = blocks; a.target_sel = -1;
    a.lut = ctx ? ctx->stab_lut : nullptr; a.reward_out = reward_out; a.pos_out = positions_out; a.stable_out = stable_out;
    a.score64_out = score64_out;
    return episode_dispatch(ctx, d, a, (hipStream_t)stream);
}
