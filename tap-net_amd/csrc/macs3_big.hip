// macs3_big.hip -- MACS / MUL 3D (tools.calc_one_position_mcs_3d, tools.py:2751-3165) for containers beyond the
// lane-per-cell kernel of tap_macs3.h: more than 64 cells or a side above 8 (e.g. --container_width 10 -> 10 x 10 x H,
// model.py:279).  ONE WAVEFRONT per container (tap_macs3_wave.h: the serial algorithm of tap_macs3_big.h run wave-uniformly on
// the container's LDS tile, its long loops shared by the lanes), on the same reduced state as the lane kernel (height-map,
// placement history, the free-list bit-grid in `occ`); one THREAD per container with the lists in the blob's scratch
// section when the tile does not fit the LDS.  gfx950 only.
#include "tap_common.h"
#include "tap_place.h"
#include "tap_macs3_big.h"
#include "tap_macs3_wave.h"
#include "tap_masks.h"
#include "tap_transition.h"

static_assert(M3B_F_HARD == TAP_F_HARD && M3B_F_USE_P == TAP_F_USE_P && M3B_F_USE_S == TAP_F_USE_S &&
              M3B_F_ZERO == TAP_F_MCS_ZERO && M3B_F_TIE == TAP_F_MCS_TIE, "flag bits are passed through");

// EMS entries one step can hold (error bit 16 beyond): the level lists contribute up to two per changed (level, row,
// run), each placed block at most four beside it and its footprint's cells on top
__host__ __device__ inline int macs3_big_cap(int n_max) { return 128 + 8 * n_max; }

// scratch ints per container: ems[cap] (2 ints each) | lev[cells] | slots[cells] | lvh[n_max + 2] | lvr[n_max + 2]
size_t tap_macs3_big_scratch_ints(const tap_env_desc *d)
{
    return (size_t)2 * macs3_big_cap(d->n_max) + (size_t)2 * d->W * d->L + (size_t)2 * (d->n_max + 2);
}

__global__ void __launch_bounds__(TAP_BLOCK) k_macs3d_big_step(StepArgs a, int32_t *scratch, size_t scratch_ints, int lpw)
{
    const int env = tap_spread_env(lpw, a.d.B);                                  // containers spread over the waves (tap_common.h)
    const int B = a.d.B, W = a.d.W, L = a.d.L, H = a.d.H, cells = W * L;
    if (env < 0) return;
    int bx, by, bz;
    if (a.static_) {                                                             // model.py:404-412
        bool badp;
        const long p = tap_col((long)a.ptr[env], a.nR, badp);
        bx = badp ? 0 : (int)a.static_[((size_t)env * a.static_rows + 1) * a.nR + p];
        by = badp ? 0 : (int)a.static_[((size_t)env * a.static_rows + 2) * a.nR + p];
        bz = badp ? 0 : (int)a.static_[((size_t)env * a.static_rows + 3) * a.nR + p];
    } else if (a.blocks_dtype == TAP_DT_F32) {
        const float *b = (const float *)a.blocks + (size_t)env * 3;
        bx = (int)b[0]; by = (int)b[1]; bz = (int)b[2];
    } else {
        const int32_t *b = (const int32_t *)a.blocks + (size_t)env * 3;
        bx = b[0]; by = b[1]; bz = b[2];
    }
    const bool act = !a.active || a.active[env] != 0;
    const int4 cv = reinterpret_cast<const int4 *>(a.v.cnt)[env];
    int cnt[4] = {cv.x, cv.y, cv.z, cv.w};
    int err = 0;
    bool do_step = act;
    if (act && cnt[3] >= a.d.n_max) { err |= 2; do_step = false; }               // tools.py:3677 IndexError
    // sides larger than the container are rejected as invalid input, as in tap_macs3.h (the reference keeps such a
    // block in its history at (0,0,0) and its later slices run out of range, tools.py:2858, 2914); footprints above
    // 8 x 8 are beyond the support mask of the stability test
    if (act && (bx < 1 || by < 1 || bz < 1 || bx > W || by > L || bx > TAP_WIDE_MAX_SIDE || by > TAP_WIDE_MAX_SIDE)) { err |= 4; do_step = false; }
    if (do_step) {
        const int step = cnt[3], cap = macs3_big_cap(a.d.n_max);
        int32_t *sc = scratch + (size_t)env * scratch_ints;
        M3BState s;
        s.W = W; s.L = L; s.H = H; s.HW = (H + 63) / 64; s.flags = a.d.flags; s.cap = cap; s.step = step;
        s.hm = a.v.hm + (size_t)env * cells;
        s.occ = a.v.occ + (size_t)env * cells * s.HW;
        s.pos = a.v.pos + env; s.blk = a.v.blk + env; s.hs = (size_t)B;
        s.ems = reinterpret_cast<M3BEms *>(sc);
        s.lev = sc + 2 * cap;
        s.slots = s.lev + cells;
        s.lvh = s.slots + cells;
        s.lvr = s.lvh + a.d.n_max + 2;
        const uint32_t *lut = a.lut;
        const M3BResult r = m3b_place(s, cnt, err, bx, by, bz,
                                      [lut](int fx, int fy, m3b_u64 eq) -> int { return tap_stable3d_any(lut, fx, fy, eq); });
        cnt[3] += 1;                                                             // tools.py:3713
        reinterpret_cast<int4 *>(a.v.cnt)[env] = make_int4(cnt[0], cnt[1], cnt[2], cnt[3]);
        a.v.pos[(size_t)(step * 3) * B + env] = r.x;
        a.v.pos[(size_t)(step * 3 + 1) * B + env] = r.y;
        a.v.pos[(size_t)(step * 3 + 2) * B + env] = r.z;
        a.v.stable[(size_t)step * B + env] = (uint8_t)r.stab;
        a.v.blk[(size_t)(step * 3) * B + env] = bx | (r.placed << 16);          // history of later steps
        a.v.blk[(size_t)(step * 3 + 1) * B + env] = by;                          // (tools.py:2843-2846), failures too
        a.v.blk[(size_t)(step * 3 + 2) * B + env] = bz;
    }
    if (err) a.v.err[env] |= err;
}

// ---- one WAVEFRONT per container (tap_macs3_wave.h): the container's working set in the wave's LDS tile ---------------
// one MACS 3D step of container `env` by one wavefront (every lane calls; env < B); base = the wave's LDS tile
__device__ __forceinline__ void macs3d_wave_body(const StepArgs &a, int env, int lane, m3b_u64 *base)
{
    const int B = a.d.B, W = a.d.W, L = a.d.L, H = a.d.H, cells = W * L, HW = (H + 63) / 64;
    const int cap = macs3_big_cap(a.d.n_max);
    M3WTile s;
    s.W = W; s.L = L; s.H = H; s.HW = HW; s.flags = a.d.flags; s.cap = cap; s.n_max = a.d.n_max;
    s.occ = base;
    s.rows = s.occ + (size_t)cells * HW;
    s.ems = reinterpret_cast<M3BEms *>(s.rows + 64);
    s.hm = reinterpret_cast<int32_t *>(s.ems + cap);
    s.lev = s.hm + cells; s.slots = s.lev + cells; s.pxy = s.slots + cells; s.bs = s.pxy + cells; s.be = s.bs + 64;
    int32_t *hpos = s.be + 64, *hblk = hpos + 3 * a.d.n_max;                    // the history so far, one round trip for all of it
    s.pos = hpos; s.blk = hblk; s.hs = 1;
    int32_t *ghm = a.v.hm + (size_t)env * cells;
    m3b_u64 *gocc = a.v.occ + (size_t)env * cells * HW;
    for (int c = lane; c < cells; c += 64) s.hm[c] = ghm[c];
    {
        const int nh = 3 * min(reinterpret_cast<const int4 *>(a.v.cnt)[env].w, a.d.n_max);
        for (int k = lane; k < nh; k += 64) { hpos[k] = a.v.pos[(size_t)k * B + env]; hblk[k] = a.v.blk[(size_t)k * B + env]; }
    }
    for (int k = lane; k < cells * HW; k += 64) s.occ[k] = gocc[k];
    int bx, by, bz;
    if (a.static_) {                                                             // model.py:404-412
        bool badp;
        const long p = tap_col((long)a.ptr[env], a.nR, badp);
        bx = badp ? 0 : (int)a.static_[((size_t)env * a.static_rows + 1) * a.nR + p];
        by = badp ? 0 : (int)a.static_[((size_t)env * a.static_rows + 2) * a.nR + p];
        bz = badp ? 0 : (int)a.static_[((size_t)env * a.static_rows + 3) * a.nR + p];
    } else if (a.blocks_dtype == TAP_DT_F32) {
        const float *b = (const float *)a.blocks + (size_t)env * 3;
        bx = (int)b[0]; by = (int)b[1]; bz = (int)b[2];
    } else {
        const int32_t *b = (const int32_t *)a.blocks + (size_t)env * 3;
        bx = b[0]; by = b[1]; bz = b[2];
    }
    const bool act = !a.active || a.active[env] != 0;
    const int4 cv = reinterpret_cast<const int4 *>(a.v.cnt)[env];
    int cnt[4] = {cv.x, cv.y, cv.z, cv.w};
    int err = 0;
    bool do_step = act;
    if (act && cnt[3] >= a.d.n_max) { err |= 2; do_step = false; }                   // tools.py:3677 IndexError
    if (act && (bx < 1 || by < 1 || bz < 1 || bx > W || by > L || bx > TAP_WIDE_MAX_SIDE || by > TAP_WIDE_MAX_SIDE)) { err |= 4; do_step = false; }   // as k_macs3d_big_step
    tap_wave_lds_sync();
    if (do_step) {                                                               // wave-uniform
        const int step = cnt[3];
        s.step = step;
        const M3BResult r = m3w_place(s, cnt, err, bx, by, bz, a.lut, lane);
        cnt[3] += 1;                                                             // tools.py:3713
        tap_wave_lds_sync();
        for (int c = lane; c < cells; c += 64) ghm[c] = s.hm[c];
        for (int k = lane; k < cells * HW; k += 64) gocc[k] = s.occ[k];
        if (lane == 0) {
            reinterpret_cast<int4 *>(a.v.cnt)[env] = make_int4(cnt[0], cnt[1], cnt[2], cnt[3]);
            a.v.pos[(size_t)(step * 3) * B + env] = r.x;
            a.v.pos[(size_t)(step * 3 + 1) * B + env] = r.y;
            a.v.pos[(size_t)(step * 3 + 2) * B + env] = r.z;
            a.v.stable[(size_t)step * B + env] = (uint8_t)r.stab;
            a.v.blk[(size_t)(step * 3) * B + env] = bx | (r.placed << 16);      // history of later steps
            a.v.blk[(size_t)(step * 3 + 1) * B + env] = by;                      // (tools.py:2843-2846), failures too
            a.v.blk[(size_t)(step * 3 + 2) * B + env] = bz;
        }
    }
    if (lane == 0 && err) a.v.err[env] |= err;
    if (a.feature_out) {                                                         // get_heightmap's feature of the new map (tools.py:3716-3744), from the tile
        tap_wave_lds_sync();
        float *o = a.feature_out + (size_t)env * a.flen;
        if (a.d.feature == TAP_FEAT_DIFF) {
            int x = lane / L, y = lane - x * L;
            const int dx = 64 / L, dy = 64 - dx * L;
            for (int c = lane; c < cells; c += 64) {
                o[c] = (float)(x > 0 ? s.hm[c] - s.hm[c - L] : 0);
                o[cells + c] = (float)(y > 0 ? s.hm[c] - s.hm[c - 1] : 0);
                x += dx; y += dy;
                if (y >= L) { y -= L; ++x; }
            }
        } else {
            int mn = 0;
            if (a.d.feature == TAP_FEAT_ZERO) {
                mn = INT_MAX;
                for (int c = lane; c < cells; c += 64) mn = min(mn, s.hm[c]);
                mn = group_min<64>(mn);
            }
            for (int c = lane; c < cells; c += 64) o[c] = (float)(s.hm[c] - mn);
        }
    }
}

__global__ void __launch_bounds__(TAP_BLOCK) k_macs3d_wave_step(StepArgs a)
{
    extern __shared__ unsigned long long m3w_lds[];
    const int lane = threadIdx.x & 63, wave_in_wg = threadIdx.x >> 6;
    const int env = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (env >= a.d.B) return;                                                     // wave-uniform
    const int cells = a.d.W * a.d.L, HW = (a.d.H + 63) / 64;
    macs3d_wave_body(a, env, lane, m3w_lds + (size_t)wave_in_wg * m3w_tile_u64(cells, HW, a.d.n_max, macs3_big_cap(a.d.n_max)));
}

// The decoding step in ONE launch (round 5): a container's wavefront runs update_dynamic + update_mask of its own
// precedence slab on the bit shadow (tap_transition.h), then its placement.
template <int NC, int MODE>
__global__ void __launch_bounds__(TAP_BLOCK) k_macs3d_wave_transition(TransArgs a, int PW, int tile_u64)
{
    extern __shared__ unsigned long long m3w_lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;   // (a vector value: TAP_WAVE_INDEX() measured slower / flat here, tap_common.h)
    const int env = blockIdx.x * PW + wave;
    if (env >= a.s.d.B) return;                                                   // wave-uniform
    // The wave first runs its container's precedence update (one slab: inputs in one round trip, write-through stores
    // that drain while the placement runs), then the placement.  Stream waves of their own, as in k_transition, would
    // occupy wave slots at this kernel's register count: with 4 + 2 waves per workgroup a CU held 8 placement waves
    // instead of 16 and the step took 218 against 148 us (MACS 3D 10 x 10, B = 4 096, round 5).
    trans_stream_wave<1, NC, MODE>(a.m, env, lane, reinterpret_cast<float *>(m3w_lds + (size_t)PW * tile_u64) + (size_t)wave * 3 * a.m.nR);
    if (lane == 0 && a.s.static_ && (a.s.dec_static_out || a.s.tour_out || a.s.picked_out)) {   // the gather's by-products
        bool badp;
        const long praw = (long)a.s.ptr[env];
        const long p = tap_col(praw, a.s.nR, badp);
        float fv[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < a.s.d.D; ++k) fv[k] = badp ? 0.f : a.s.static_[((size_t)env * a.s.static_rows + 1 + k) * a.s.nR + p];
        tap_step_aux(a.s, env, a.s.d.D, fv, praw);
    }
    macs3d_wave_body(a.s, env, lane, m3w_lds + (size_t)wave * tile_u64);
}

static int macs3d_transition_pw(const tap_ctx *ctx, const tap_env_desc *d, int nR)
{
    if (tap_wave_kernels_off()) return 0;
    const size_t tile = m3w_tile_u64(d->W * d->L, (d->H + 63) / 64, d->n_max, macs3_big_cap(d->n_max)) * 8;
    for (int pw = 4; pw >= 1; pw >>= 1)
        if ((size_t)pw * tile + (size_t)pw * 3 * nR * sizeof(float) <= tap_lds_limit(ctx)) return pw;
    return 0;
}

bool tap_macs3_wave_transition_ok(const tap_ctx *ctx, const tap_env_desc *d, int nR) { return macs3d_transition_pw(ctx, d, nR) > 0; }

int tap_macs3_wave_transition(tap_ctx *ctx, const tap_env_desc *d, const TransArgs &a, hipStream_t st)
{
    const int pw = macs3d_transition_pw(ctx, d, a.m.nR);
    if (pw == 0) return tap_fail(ctx, TAP_E_UNSUPPORTED, "no fused step for this container");
    if (!a.s.v.scratch || !a.s.v.occ) return tap_fail(ctx, TAP_E_INVALID, "MACS 3D above 64 cells: the state blob has no scratch section");
    const int tile_u64 = (int)m3w_tile_u64(d->W * d->L, (d->H + 63) / 64, d->n_max, macs3_big_cap(d->n_max));
    const int mode = a.m.bits_in ? 1 : 2;
    const size_t lds = (size_t)pw * tile_u64 * 8 + (size_t)pw * 3 * a.m.nR * sizeof(float);
    const dim3 g((d->B + pw - 1) / pw), blk(64 * pw);
    if (g.x == 0) return TAP_OK;
#define TAP_MT(NC_, M_) do { TAP_HIP_CHECK(ctx, tap_allow_lds(k_macs3d_wave_transition<NC_, M_>, lds)); \
        hipLaunchKernelGGL((k_macs3d_wave_transition<NC_, M_>), g, blk, lds, st, a, pw, tile_u64); } while (0)
#define TAP_MT_M(NC_) do { if (mode == 1) TAP_MT(NC_, 1); else TAP_MT(NC_, 2); } while (0)
    switch (mask_fast_path_cols(a.m)) { case 1: TAP_MT_M(1); break; case 2: TAP_MT_M(2); break; default: TAP_MT_M(4); break; }
#undef TAP_MT_M
#undef TAP_MT
    TAP_LAUNCH_CHECK(ctx, "k_macs3d_wave_transition");
    return TAP_OK;
}

int tap_macs3_big_step(tap_ctx *ctx, const StepArgs &a, hipStream_t st)
{
    if (a.d.B == 0) return TAP_OK;
    if (!a.v.scratch || !a.v.occ) return tap_fail(ctx, TAP_E_INVALID, "MACS 3D above 64 cells: the state blob has no scratch section");
    {   // one wavefront per container when its working set fits a wave's share of the LDS
        const int cells = a.d.W * a.d.L, HW = (a.d.H + 63) / 64;
        const size_t tile = m3w_tile_u64(cells, HW, a.d.n_max, macs3_big_cap(a.d.n_max)) * 8;
        int waves = TAP_BLOCK / 64;
        while (waves > 1 && (size_t)waves * tile > tap_lds_limit(ctx)) waves >>= 1;
        if ((size_t)waves * tile <= tap_lds_limit(ctx) && !tap_wave_kernels_off()) {
            TAP_HIP_CHECK(ctx, tap_allow_lds(k_macs3d_wave_step, (size_t)waves * tile));
            hipLaunchKernelGGL(k_macs3d_wave_step, dim3((a.d.B + waves - 1) / waves), dim3(waves * 64), (size_t)waves * tile, st, a);
            TAP_LAUNCH_CHECK(ctx, "k_macs3d_wave_step");                            // (writes the feature itself)
            return TAP_OK;
        }
    }
    const int lpw = tap_spread_lpw(a.d.B);                                     // containers per wavefront (tap_common.h)
    hipLaunchKernelGGL(k_macs3d_big_step, dim3(tap_spread_grid(a.d.B, lpw, TAP_BLOCK)), dim3(TAP_BLOCK), 0, st, a, a.v.scratch,
                       tap_macs3_big_scratch_ints(&a.d), lpw);
    TAP_LAUNCH_CHECK(ctx, "k_macs3d_big_step");
    if (a.feature_out) return tap_big_feature(ctx, &a.d, a.v, a.feature_out, a.flen, st);   // tools.py:3716-3744
    return TAP_OK;
}

#ifdef M3W_PROF
extern "C" int tap_m3w_prof_read(unsigned long long *out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(m3w_prof), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(m3w_prof), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif
