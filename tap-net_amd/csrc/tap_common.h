// tap_common.h -- host-side plumbing shared by the libtapenv translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "tapenv.h"

constexpr int TAP_CHK_SLOTS = 64;
constexpr int TAP_BLOCK = 256; // threads per workgroup (4 wave64)

struct tap_ctx {
    int device;
    uint32_t *stab_lut; // device: tap_stable3d for footprints <= 4x4 (tap_place.h), built at create
    int32_t *chk;       // device: TAP_CHK_SLOTS x 2 ints (flagged envs, OR of their error words) for tap_env_check
    unsigned chk_next;  //   next slot
    size_t lds_limit;   // LDS a workgroup may allocate on this device (gfx950: 160 KiB per CU), queried at create
    char err[512];
};

// LDS budget of one workgroup: what the device reports (hipDeviceAttributeMaxSharedMemoryPerBlock; 160 KiB on
// gfx950), never the 64 KiB of earlier CDNA parts
inline size_t tap_lds_limit(const tap_ctx *ctx) { return (ctx && ctx->lds_limit) ? ctx->lds_limit : (size_t)64 * 1024; }

// Threads per workgroup for kernels that keep `per_env` bytes of dynamic LDS per container (G lanes each): the
// containers resident on a CU are bounded by LDS / per_env whatever the workgroup size, so a workgroup is sized to
// leave room for a second one (<= limit / 2) where that is possible -- smaller workgroups pack the CU's LDS better --
// and only a container that needs more than half the LDS gets a workgroup to itself.  0 = does not fit at all.
inline int tap_lds_threads(size_t per_env, int G, size_t limit, size_t static_bytes = 0)
{
    const size_t budget = limit > static_bytes ? limit - static_bytes : 0;
    int threads = TAP_BLOCK;
    while (threads > 64 && (size_t)(threads / G) * per_env > budget / 2) threads /= 2;
    if (threads < G) threads = G;
    while (threads > G && threads > 64 && (size_t)(threads / G) * per_env > budget) threads /= 2;
    return (size_t)(threads / G) * per_env <= budget ? threads : 0;
}

// a launch whose dynamic LDS exceeds the 64 KiB default window must raise the kernel's limit first
template <typename K> inline hipError_t tap_allow_lds(K kernel, size_t bytes)
{
    if (bytes <= (size_t)64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

inline int tap_fail(tap_ctx *ctx, int code, const char *fmt, ...)
{
    if (ctx) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

#define TAP_HIP_CHECK(ctx, expr)                                                             \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return tap_fail(ctx, TAP_E_HIP, "%s failed: %s (%s:%d)", #expr,                  \
                            hipGetErrorString(e_), __FILE__, __LINE__);                      \
    } while (0)

#define TAP_LAUNCH_CHECK(ctx, what)                                                          \
    do {                                                                                     \
        hipError_t e_ = hipGetLastError();                                                   \
        if (e_ != hipSuccess)                                                                \
            return tap_fail(ctx, TAP_E_HIP, "launch of %s failed: %s", what,                 \
                            hipGetErrorString(e_));                                          \
    } while (0)


// A column index handed in by the caller (ptr / tour).  The reference raises IndexError for a value
// outside [0, nR); here it must never become an out-of-bounds read: the index is replaced by column 0
// and `bad` makes the caller flag the step (error bit 4, "bad block").
__device__ __forceinline__ long tap_col(long p, int nR, bool &bad)
{
    bad = p < 0 || p >= nR;
    return bad ? 0 : p;
}


// ---- state blob layout -------------------------------------------------------------------
// hm      int32 [B][cells]        height-map, env-major (a lane group reads one env's row)
// cnt     int32 [B][4]            valid_size, empty_size, sum(stable), current_blocks_num
// err     int32 [B]               sticky error bits (1 = overflow, 2 = too many steps)
// pos     int32 [n_max*D][B]      positions, step-major so one lock-step is one coalesced row
// stable  uint8 [n_max][B]
// blk     int32 [n_max*D][B]      placed block sizes (MACS history only)
// occ     uint64 [B][cells][HW]   MACS 3D only: complement of the cell's free-list column, HW = ceil(H/64)
//                                 words (tap_macs3.h)
struct EnvView {
    int32_t *hm;
    int32_t *cnt;
    int32_t *err;
    int32_t *pos;
    uint8_t *stable;
    int32_t *blk;
    unsigned long long *occ;
    int32_t *scratch; // LB_GREEDY above 64 cells (big.hip): cells ints per container; MACS 2D above 64 columns (macs_big.hip); MACS 3D above 64 cells (macs3_big.hip)
    int16_t *vox;     // legacy LB (lb.hip): [B][cells][H] block ids, -1 under covered holes   (tools.py:3630)
    uint8_t *lfs;     //   [B][H*L][W+2] level_free_space lists                                (tools.py:3649-3653)
    uint8_t *lfn;     //   [B][H*L] (list length - 1) mod 256: the zeroed blob is the initial [0] everywhere
};

inline size_t tap_align256(size_t x) { return (x + 255) & ~size_t(255); }

// LB_GREEDY containers beyond the lane-per-cell kernels (more than 64 cells, or a 3D side above 8): big.hip
inline bool tap_is_big(const tap_env_desc *d)
{
    return d->strategy == TAP_LB_GREEDY && (d->W * d->L > 64 || (d->D == 3 && (d->W > 8 || d->L > 8)));
}

// MACS / MUL 2D containers beyond the lane-per-column kernels (more than 64 columns): macs_big.hip, one thread per
// container with its candidate lists in a scratch section of the blob
inline bool tap_is_big_macs(const tap_env_desc *d) { return d->strategy == TAP_MACS && d->D == 2 && d->W > 64; }
size_t tap_macs_big_scratch_ints(const tap_env_desc *d);   // macs_big.hip
// MACS / MUL 3D containers beyond the lane-per-cell kernel (more than 64 cells or a side above 8): macs3_big.hip, one
// thread per container on the same state (height-map, history, free-list bit-grid) + candidate lists in the scratch section
inline bool tap_is_big_macs3(const tap_env_desc *d)
{
    return d->strategy == TAP_MACS && d->D == 3 && (d->W * d->L > 64 || d->W > 8 || d->L > 8);
}
size_t tap_macs3_big_scratch_ints(const tap_env_desc *d);  // macs3_big.hip

inline size_t tap_env_layout(const tap_env_desc *d, void *base, EnvView *v)
{
    const size_t B = (size_t)d->B, cells = (size_t)d->W * d->L, nD = (size_t)d->n_max * d->D;
    size_t off = 0;
    char *p = static_cast<char *>(base);
    auto take = [&](size_t bytes) { size_t o = off; off = tap_align256(off + bytes); return o; };
    size_t o_hm = take(B * cells * 4), o_cnt = take(B * 16), o_err = take(B * 4);
    size_t o_pos = take(nD * B * 4), o_st = take((size_t)d->n_max * B);
    size_t o_blk = (d->strategy == TAP_MACS || d->strategy == TAP_LB) ? take(nD * B * 4) : 0;
    size_t o_occ = (d->strategy == TAP_MACS && d->D == 3) ? take(B * cells * 8 * (size_t)((d->H + 63) / 64)) : 0;
    const bool lb = d->strategy == TAP_LB;
    const bool scr = tap_is_big(d) || tap_is_big_macs(d) || tap_is_big_macs3(d);
    size_t o_scr = tap_is_big(d) ? take(B * cells * 4) : tap_is_big_macs(d) ? take(B * tap_macs_big_scratch_ints(d) * 4)
                 : tap_is_big_macs3(d) ? take(B * tap_macs3_big_scratch_ints(d) * 4) : 0;
    size_t o_vox = lb ? take(B * cells * (size_t)d->H * 2) : 0;
    size_t o_lfs = lb ? take(B * (size_t)d->H * d->L * (size_t)(d->W + 2)) : 0;
    size_t o_lfn = lb ? take(B * (size_t)d->H * d->L) : 0;
    if (v) {
        v->scratch = scr ? reinterpret_cast<int32_t *>(p + o_scr) : nullptr;
        v->vox = lb ? reinterpret_cast<int16_t *>(p + o_vox) : nullptr;
        v->lfs = lb ? reinterpret_cast<uint8_t *>(p + o_lfs) : nullptr;
        v->lfn = lb ? reinterpret_cast<uint8_t *>(p + o_lfn) : nullptr;
        v->hm = reinterpret_cast<int32_t *>(p + o_hm);
        v->cnt = reinterpret_cast<int32_t *>(p + o_cnt);
        v->err = reinterpret_cast<int32_t *>(p + o_err);
        v->pos = reinterpret_cast<int32_t *>(p + o_pos);
        v->stable = reinterpret_cast<uint8_t *>(p + o_st);
        v->blk = (d->strategy == TAP_MACS || d->strategy == TAP_LB) ? reinterpret_cast<int32_t *>(p + o_blk) : nullptr;
        v->occ = (d->strategy == TAP_MACS && d->D == 3) ? reinterpret_cast<unsigned long long *>(p + o_occ) : nullptr;
    }
    return off;
}

// arguments of one lock-step placement launch (env.hip, macs.hip)
struct StepArgs {
    tap_env_desc d;
    EnvView v;
    const void *blocks;   // (B, D) f32 | i32, or null when gathering
    int blocks_dtype;
    const float *static_; // gather source (B, static_rows, nR)
    int static_rows, nR;
    const int64_t *ptr;
    const uint8_t *active;
    float *feature_out;
    int flen;
    const uint32_t *lut; // ctx->stab_lut
    // by-products of the gather for a caller that drives the decoding loop (stepper.hip; model.py:404-406, 512):
    // the chosen block's sides as the floats `static` holds -- decoder_static (B, D, 1) -- and the pick itself
    // into column tour_col of a (B, tour_stride) int64 tour.  Null = not wanted.  Written by the fused kernels'
    // placement waves (tap_step_aux); the stand-alone steps ignore them.
    float *dec_static_out;
    int64_t *tour_out;
    int tour_stride, tour_col;
    // rolling windows (rolling.py:637 sub_graph_nodes[ptr]): the global id of the picked block -- entry ptr % child of
    // the CURRENT window's node list (B, child) -- into column tour_col of a (B, tour_stride) int32 array
    const int32_t *nodes_cur;
    int32_t *picked_out;
    int child;
};

// model.py:404-406 (decoder_static = static[:, 1:, ptr]) and the tour column of model.py:512, by the lane that
// holds the gathered sides; v = the D floats read from `static` (0 for an index outside [0, nR))
__device__ __forceinline__ void tap_step_aux(const StepArgs &s, int env, int D, const float *v, long ptr_raw)
{
    if (s.dec_static_out)
        for (int k = 0; k < D; ++k) s.dec_static_out[(size_t)env * D + k] = v[k];
    if (s.tour_out) s.tour_out[(size_t)env * s.tour_stride + s.tour_col] = (int64_t)ptr_raw;
    if (s.picked_out) {
        const bool ok = ptr_raw >= 0 && ptr_raw < s.nR;
        s.picked_out[(size_t)env * s.tour_stride + s.tour_col] = ok ? s.nodes_cur[(size_t)env * s.child + (int)(ptr_raw % s.child)] : -1;
    }
}

// One-THREAD-per-container kernels (macs_big.hip, macs3_big.hip, lb.hip, big.hip's fallback): a wavefront runs the union of
// its lanes' control flows, and these placements are long, data-dependent serial loops -- 64 containers per wave cost
// about 64 times one.  So the containers are SPREAD: only the first `lpw` lanes of every wave carry a container, with
// lpw chosen so that the launch has about four waves per SIMD (B = 4 096 -> one container per wave).  env of this thread,
// or -1 for an idle lane:
__device__ __forceinline__ int tap_spread_env(int lpw, int B)
{
    const int lane = threadIdx.x & 63, wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int env = wave * lpw + lane;
    return (lane < lpw && env < B) ? env : -1;
}
// A wavefront's index inside its workgroup, in a SCALAR register.  threadIdx.x >> 6 is the same on every lane of a wave,
// but the compiler only sees a vector value: for kernels in which a wave owns one unit (a container, an instance, a pair of
// slabs) everything derived from the index -- the unit's number, every address -- then runs on the vector ALU.  Through
// readfirstlane it is scalar arithmetic + lane offsets.  Measured per kernel family (round 6, profiles/r06_stream_wave_ab.txt,
// r06_rolling_swave_ab.txt, r06_wave_index_big_ab.txt): the fused step with two slabs per stream wave (c2) + 4 %, the
// rolling step (c5) + 6 %, MACS 2D one wavefront per container (c9) + 5 % -- used there; one slab per stream wave (c3)
// - 0.7 %, the MACS lane kernels (c4, c6) and MACS 3D one wavefront per container (c8) flat, LB_GREEDY one wavefront per
// container (c7) - 4 % -- not used there.  -DTAP_WAVE_INDEX_VECTOR for A/B builds.
#ifdef TAP_WAVE_INDEX_VECTOR
#define TAP_WAVE_INDEX() ((int)(threadIdx.x >> 6))
#else
#define TAP_WAVE_INDEX() (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)))
#endif

// TAP_NO_WAVE_KERNELS (read once): the wave-per-container kernels of big.hip / macs_big.hip / macs3_big.hip stand aside
// and their fallbacks -- what runs when a container's tile does not fit the LDS -- take every launch (parity tests, A/B)
inline bool tap_wave_kernels_off() { static const bool off = getenv("TAP_NO_WAVE_KERNELS") != nullptr; return off; }
inline int tap_spread_lpw(int B)                 // lanes per wave that carry a container
{
    int lpw = 64;
    while (lpw > 1 && (B + lpw - 1) / lpw < 4096) lpw >>= 1;        // 256 CUs x 4 SIMDs x 4 waves
    return lpw;
}
inline int tap_spread_grid(int B, int lpw, int threads) { const int waves = (B + lpw - 1) / lpw, wpb = threads / 64; return (waves + wpb - 1) / wpb; }

int tap_desc_validate(tap_ctx *ctx, const tap_env_desc *d);
// lanes per env: smallest of 8/16/32/64 that holds W*L cells, 0 if unsupported
inline int tap_group_size(const tap_env_desc *d)
{
    const int cells = d->W * d->L;
    static const int forced = [] { const char *e = getenv("TAP_FORCE_G"); return e ? atoi(e) : 0; }();   // A/B runs only
    if (forced >= cells && (forced == 8 || forced == 16 || forced == 32 || forced == 64)) return forced;
    return cells <= 8 ? 8 : cells <= 16 ? 16 : cells <= 32 ? 32 : cells <= 64 ? 64 : 0;
}
// the same by-products from a launch of their own, for the steps that run as two launches (transition.hip)
int tap_step_aux_launch(tap_ctx *ctx, const StepArgs &s, hipStream_t st);
int tap_macs_big_step(tap_ctx *ctx, const StepArgs &a, hipStream_t st);                                             // macs_big.hip
int tap_macs_wave_step(tap_ctx *ctx, const StepArgs &a, hipStream_t st);                                            // macs_big.hip
int tap_macs3_big_step(tap_ctx *ctx, const StepArgs &a, hipStream_t st);                                            // macs3_big.hip
int tap_big_step(tap_ctx *ctx, const StepArgs &a, void *state, hipStream_t st);                                      // big.hip
int tap_big_feature(tap_ctx *ctx, const tap_env_desc *d, const EnvView &v, float *out, int flen, hipStream_t st);   // big.hip

// dispatch on (D, lanes per container) for the lane-per-cell kernels
#define TAP_DISPATCH_DG(fn, d, ...)                                                          \
    do {                                                                                     \
        const int G_ = tap_group_size(d);                                                    \
        if ((d)->D == 2) {                                                                   \
            if (G_ == 8) return fn<2, 8>(__VA_ARGS__);                                       \
            if (G_ == 16) return fn<2, 16>(__VA_ARGS__);                                     \
            if (G_ == 32) return fn<2, 32>(__VA_ARGS__);                                     \
            return fn<2, 64>(__VA_ARGS__);                                                   \
        } else {                                                                             \
            if (G_ == 8) return fn<3, 8>(__VA_ARGS__);                                       \
            if (G_ == 16) return fn<3, 16>(__VA_ARGS__);                                     \
            if (G_ == 32) return fn<3, 32>(__VA_ARGS__);                                     \
            return fn<3, 64>(__VA_ARGS__);                                                   \
        }                                                                                    \
    } while (0)
