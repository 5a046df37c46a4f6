// env.hip -- tools.Container for B containers in lock-step: context, descriptor, reset, step,
// feature, ratio, export, check (the whole-episode kernels are episode.hip).  gfx950 only.
#include "tap_common.h"
#include "tap_place.h"
#include "tap_waves.h"

// =============================================================================================
// context / descriptor (host)
// =============================================================================================

extern "C" int tap_abi_version(void) { return TAP_ABI_VERSION; }

extern "C" const char *tap_status_string(int s)
{
    switch (s) {
    case TAP_OK: return "ok";
    case TAP_E_INVALID: return "invalid argument";
    case TAP_E_UNSUPPORTED: return "unsupported configuration";
    case TAP_E_HIP: return "HIP runtime error";
    case TAP_E_OVERFLOW: return "placement above container height";
    case TAP_E_NODEVICE: return "no HIP device";
    case TAP_E_STEPS: return "more add_new_block calls than blocks_num";
    default: return "unknown status";
    }
}

// one thread per (shape, packed mask): evaluate the hull-free predicate, set the bit (tap_place.h: layout)
__global__ void k_build_stab_lut(uint32_t *lut)
{
    const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (unsigned)TAP_LUT_BITS) return;
    const int shape = (int)(idx >> 16), bx = shape / 4 + 1, by = shape % 4 + 1;
    const unsigned packed = idx & 0xffffu;
    for (int i = 0; i < 4; ++i) {                              // only masks inside the bx x by footprint occur
        const unsigned row = (packed >> (4 * i)) & 15u;
        if (row && (i >= bx || (row >> by))) return;
    }
    if (tap_stable3d(bx, by, tap_lut_unpack(packed))) atomicOr(&lut[idx >> 5], 1u << (idx & 31u));
}

extern "C" int tap_ctx_create(int device, tap_ctx **out)
{
    if (!out) return TAP_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return TAP_E_NODEVICE;
    if (device < 0 || device >= n) return TAP_E_INVALID;
    tap_ctx *c = new tap_ctx();
    c->device = device;
    c->err[0] = 0;
    c->stab_lut = nullptr;
    int prev = 0;
    (void)hipGetDevice(&prev);
    c->chk = nullptr;
    c->chk_next = 0;
    int lds = 0;
    c->lds_limit = (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device) == hipSuccess && lds > 0)
                       ? (size_t)lds : (size_t)64 * 1024;
    bool ok = hipSetDevice(device) == hipSuccess &&
              hipMalloc(reinterpret_cast<void **>(&c->chk), 2 * TAP_CHK_SLOTS * sizeof(int32_t)) == hipSuccess &&
              hipMalloc(reinterpret_cast<void **>(&c->stab_lut), TAP_LUT_WORDS * sizeof(uint32_t)) == hipSuccess &&
              hipMemset(c->stab_lut, 0, TAP_LUT_WORDS * sizeof(uint32_t)) == hipSuccess;
    if (ok) {
        hipLaunchKernelGGL(k_build_stab_lut, dim3((TAP_LUT_BITS + 255) / 256), dim3(256), 0, 0, c->stab_lut);
        ok = hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess;
    }
    (void)hipSetDevice(prev);
    if (!ok) {
        if (c->chk) (void)hipFree(c->chk);
        if (c->stab_lut) (void)hipFree(c->stab_lut);
        delete c;
        return TAP_E_HIP;
    }
    *out = c;
    return TAP_OK;
}

extern "C" void tap_ctx_destroy(tap_ctx *ctx)
{
    if (!ctx) return;
    if (ctx->stab_lut) (void)hipFree(ctx->stab_lut);
    if (ctx->chk) (void)hipFree(ctx->chk);
    delete ctx;
}

// tools.is_stable (tools.py:710-765) on explicit support masks -- see tapenv.h
__global__ void __launch_bounds__(TAP_BLOCK) k_stable3d_eval(int bx, int by, const unsigned long long *masks, int n,
                                                             const uint32_t *lut, uint8_t *out)
{
    const int i = blockIdx.x * TAP_BLOCK + threadIdx.x;
    if (i >= n) return;
    const u64 mc = masks[i];
    u64 m = 0; // row-major (i*by + j) -> the kernels' stride-8 form
    for (int r = 0; r < bx; ++r) m |= ((mc >> (r * by)) & ((1ull << by) - 1ull)) << (8 * r);
    out[i] = (uint8_t)(lut ? tap_stable3d_any(lut, bx, by, m) : tap_stable3d(bx, by, m));
}

extern "C" int tap_stable3d_eval(tap_ctx *ctx, int bx, int by, const unsigned long long *masks, int n,
                                 int use_lut, uint8_t *stable_out, void *stream)
{
    if (!ctx) return TAP_E_INVALID;
    if (bx < 1 || by < 1 || bx > 8 || by > 8 || n < 0)
        return tap_fail(ctx, TAP_E_INVALID, "stable3d_eval: footprint sides must be 1..8");
    if (n == 0) return TAP_OK;
    if (!masks || !stable_out) return tap_fail(ctx, TAP_E_INVALID, "null pointer");
    hipLaunchKernelGGL(k_stable3d_eval, dim3((n + TAP_BLOCK - 1) / TAP_BLOCK), dim3(TAP_BLOCK), 0, (hipStream_t)stream,
                       bx, by, masks, n, use_lut ? ctx->stab_lut : nullptr, stable_out);
    TAP_LAUNCH_CHECK(ctx, "k_stable3d_eval");
    return TAP_OK;
}

extern "C" const char *tap_last_error(const tap_ctx *ctx) { return ctx ? ctx->err : ""; }

static bool str_ends(const char *s, const char *suf)
{
    size_t a = strlen(s), b = strlen(suf);
    return a >= b && strcmp(s + a - b, suf) == 0;
}

extern "C" int tap_env_desc_init(tap_env_desc *d, int B, int D, const int32_t *cs, int blocks_num,
                                 const char *reward, const char *hm_type, const char *strategy)
{
    if (!d || !cs || !reward || !hm_type || !strategy) return TAP_E_INVALID;
    if (D != 2 && D != 3) return TAP_E_INVALID;
    memset(d, 0, sizeof(*d));
    d->B = B; d->D = D;
    d->W = cs[0]; d->L = D == 3 ? cs[1] : 1; d->H = cs[D - 1];
    d->n_max = blocks_num;
    // tools.py:3617-3620: the reward string overrides the strategy
    const char *st = strategy;
    if (!strcmp(reward, "C+P+S-mul-soft") || !strcmp(reward, "C+P+S-mul-hard")) st = "MUL";
    else if (!strcmp(reward, "C+P+S-mcs-soft") || !strcmp(reward, "C+P+S-mcs-hard")) st = "MACS";
    if (!strcmp(st, "MACS") || !strcmp(st, "MUL")) d->strategy = TAP_MACS;
    else if (!strcmp(st, "LB_GREEDY")) d->strategy = TAP_LB_GREEDY;
    else if (!strcmp(st, "LB")) d->strategy = TAP_LB;              // tools.py:3683-3686 (lb.hip)
    else return TAP_E_UNSUPPORTED; // the pack-net back-ends are out of scope
    if (str_ends(reward, "hard")) d->flags |= TAP_F_HARD;          // tools.py:2113
    if (strchr(reward, 'P')) d->flags |= TAP_F_USE_P;              // tools.py:2135
    if (strchr(reward, 'S')) d->flags |= TAP_F_USE_S;              // tools.py:2138
    if (!strncmp(reward, "mcs", 3)) d->flags |= TAP_F_MCS_ZERO;    // tools.py:2709
    if (strstr(reward, "mcs")) d->flags |= TAP_F_MCS_TIE;          // tools.py:2718
#ifdef TAP_AB_BUILD
    // A/B builds only (-DTAP_AB_BUILD, scripts/ab_transition.sh; never in the product library, whose results no
    // environment variable can change): TAP_AB_NO_TIE=1 runs 'mcs' without its usable-space tie-break -- the placements
    // then differ from the reference's -- to bound what that phase costs inside a launch
    if (getenv("TAP_AB_NO_TIE")) {
        static bool warned = false;
        if (!warned) { fprintf(stderr, "libtapenv (A/B build): TAP_AB_NO_TIE is set -- 'mcs' placements differ from the reference's\n"); warned = true; }
        d->flags &= ~TAP_F_MCS_TIE;
    }
#endif
    // tools.py:3919-3964
    struct { const char *name; int mode; } table[] = {
        {"comp", TAP_R_C}, {"soft", TAP_R_CxS}, {"hard", TAP_R_CxS}, {"pyrm", TAP_R_CP},
        {"pyrm-soft", TAP_R_CPxS}, {"pyrm-hard", TAP_R_CPxS}, {"mcs-soft", TAP_R_CPxS},
        {"mcs-hard", TAP_R_CPxS}, {"pyrm-soft-sum", TAP_R_CPS}, {"pyrm-soft-SUM", TAP_R_2CPS},
        {"pyrm-hard-sum", TAP_R_CPS}, {"pyrm-hard-SUM", TAP_R_2CPS}, {"CPS", TAP_R_CxPxS},
        {"C+P-lb-soft", TAP_R_CP_HALF}};
    d->ratio_mode = TAP_R_CPS;
    for (auto &t : table) if (!strcmp(reward, t.name)) d->ratio_mode = t.mode;
    if (!strcmp(hm_type, "full")) d->feature = TAP_FEAT_FULL;
    else if (!strcmp(hm_type, "zero")) d->feature = TAP_FEAT_ZERO;
    else if (!strcmp(hm_type, "diff")) d->feature = TAP_FEAT_DIFF;
    else return TAP_E_INVALID;
    return TAP_OK;
}

int tap_desc_validate(tap_ctx *ctx, const tap_env_desc *d)
{
    if (!d) return tap_fail(ctx, TAP_E_INVALID, "null descriptor");
    if ((d->D != 2 && d->D != 3) || d->B < 0 || d->W < 1 || d->L < 1 || d->H < 1 || d->n_max < 1 ||
        (d->D == 2 && d->L != 1))
        return tap_fail(ctx, TAP_E_INVALID, "bad descriptor B=%d D=%d W=%d L=%d H=%d n=%d", d->B,
                        d->D, d->W, d->L, d->H, d->n_max);
    if (tap_is_big(d)) {                                           // one wavefront per container (big.hip)
        if (d->W * d->L > 4096) return tap_fail(ctx, TAP_E_UNSUPPORTED, "W*L = %d cells > 4096", d->W * d->L);
    } else if (d->strategy == TAP_LB) {                            // one thread per container (lb.hip)
        if (d->W > 248 || (d->D == 3 && d->L > 248)) return tap_fail(ctx, TAP_E_UNSUPPORTED, "legacy LB: side > 248");
    } else if (tap_is_big_macs(d)) {                               // one wavefront per container (macs_big.hip)
        if (d->W > 4096) return tap_fail(ctx, TAP_E_UNSUPPORTED, "MACS / MUL 2D: W = %d columns > 4096", d->W);
    } else if (tap_is_big_macs3(d)) {                              // one wavefront per container (macs3_big.hip)
        if (d->W > 64 || d->L > 64) return tap_fail(ctx, TAP_E_UNSUPPORTED, "MACS / MUL 3D: side > 64 (rows are 64-bit masks)");
    } else if (tap_group_size(d) == 0) {
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "W*L = %d cells > 64 lanes per container", d->W * d->L);
    }
    if (d->H > 4000 || d->n_max > 4096)
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "H or blocks_num too large for the 32-bit sort key");
    if (d->strategy != TAP_LB_GREEDY && d->strategy != TAP_MACS && d->strategy != TAP_LB)
        return tap_fail(ctx, TAP_E_INVALID, "bad strategy %d", d->strategy);
    return TAP_OK;
}

extern "C" size_t tap_env_state_bytes(const tap_env_desc *d)
{
    if (!d || d->B < 0 || d->W < 1 || d->L < 1 || d->n_max < 1) return 0;
    return tap_env_layout(d, nullptr, nullptr);
}

extern "C" int tap_env_feature_len(const tap_env_desc *d)
{
    if (!d) return 0;
    if (d->feature == TAP_FEAT_DIFF) return d->D == 2 ? d->W - 1 : 2 * d->W * d->L;
    return d->W * d->L;
}

// clear_container as a KERNEL, not hipMemsetAsync: a memset captured into a hipGraph becomes a memset node, and
// on this ROCm stack (7.2) such a node was observed to run out of order with the kernel nodes around it when the
// graph is replayed (rolling passes replayed from a graph started some episodes on a half-cleared container);
// a kernel node keeps stream order.
__global__ void __launch_bounds__(TAP_BLOCK) k_zero16(uint4 *p, size_t n16)
{
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (size_t i = (size_t)blockIdx.x * TAP_BLOCK + threadIdx.x; i < n16; i += (size_t)gridDim.x * TAP_BLOCK) p[i] = z;
}

extern "C" int tap_env_reset(tap_ctx *ctx, const tap_env_desc *d, void *state, void *stream)
{
    int rc = tap_desc_validate(ctx, d);
    if (rc) return rc;
    if (d->B == 0) return TAP_OK; // an empty batch has no buffers to check
    if (!state) return tap_fail(ctx, TAP_E_INVALID, "null state");
    const size_t bytes = tap_env_layout(d, nullptr, nullptr);       // a multiple of 256
    if (reinterpret_cast<uintptr_t>(state) % 16)
        // every section of the blob is laid out for 16-byte accesses; torch allocations are 256-byte aligned.  (No
        // hipMemsetAsync fallback: captured into a hipGraph it becomes the memset node that replays out of order.)
        return tap_fail(ctx, TAP_E_INVALID, "the state blob must be 16-byte aligned");
    const size_t n16 = bytes / 16;
    const unsigned grid = (unsigned)((n16 + TAP_BLOCK - 1) / TAP_BLOCK < 2048 ? (n16 + TAP_BLOCK - 1) / TAP_BLOCK : 2048);
    hipLaunchKernelGGL(k_zero16, dim3(grid ? grid : 1), dim3(TAP_BLOCK), 0, (hipStream_t)stream,
                       reinterpret_cast<uint4 *>(state), n16);
    TAP_LAUNCH_CHECK(ctx, "k_zero16");
    return TAP_OK;
}

// =============================================================================================
// K1/K2: one lock-step placement for B containers
// =============================================================================================

template <int D, int G>
__global__ void __launch_bounds__(TAP_BLOCK) k_env_step(StepArgs a)
{
    // every wave is a placement wave (tap_waves.h); lane groups never span a wave, so there is no
    // workgroup barrier
    __shared__ int s_old[TAP_BLOCK];
    __shared__ int s_new[TAP_BLOCK];
    const int tid = threadIdx.x, cell = tid % G;
    tap_lb_place_wave<D, G>(a, 0, nullptr, blockIdx.x * (TAP_BLOCK / G) + tid / G, cell, tid & 63,
                            s_old + (tid - cell), s_new + (tid - cell));
}

template <int D, int G> static int launch_step(tap_ctx *ctx, const StepArgs &a, hipStream_t st)
{
    const int epb = TAP_BLOCK / G;
    const int grid = (a.d.B + epb - 1) / epb;
    if (grid == 0) return TAP_OK;
    hipLaunchKernelGGL((k_env_step<D, G>), dim3(grid), dim3(TAP_BLOCK), 0, st, a);
    TAP_LAUNCH_CHECK(ctx, "k_env_step");
    return TAP_OK;
}


int tap_macs2d_step(tap_ctx *ctx, const StepArgs &a, hipStream_t st); // macs.hip
int tap_lb_step(tap_ctx *ctx, const StepArgs &a, void *state, hipStream_t st); // lb.hip

static int step_common(tap_ctx *ctx, const tap_env_desc *d, void *state, StepArgs &a, void *stream)
{
    int rc = tap_desc_validate(ctx, d);
    if (rc) return rc;
    if (d->B == 0) return TAP_OK; // an empty batch has no buffers to check
    if (!state) return tap_fail(ctx, TAP_E_INVALID, "null state");
    a.d = *d;
    tap_env_layout(d, state, &a.v);
    a.flen = tap_env_feature_len(d);
    a.lut = ctx ? ctx->stab_lut : nullptr;
    if (d->strategy == TAP_MACS) return tap_macs2d_step(ctx, a, (hipStream_t)stream);
    if (d->strategy == TAP_LB) return tap_lb_step(ctx, a, state, (hipStream_t)stream);
    if (tap_is_big(d)) return tap_big_step(ctx, a, state, (hipStream_t)stream);
    TAP_DISPATCH_DG(launch_step, d, ctx, a, (hipStream_t)stream);
}

extern "C" int tap_env_step(tap_ctx *ctx, const tap_env_desc *d, void *state, const void *blocks,
                            int blocks_dtype, const uint8_t *active, float *feature_out,
                            void *stream)
{
    if (d && d->B == 0) return tap_desc_validate(ctx, d); // an empty batch has no buffers to check
    if (!blocks) return tap_fail(ctx, TAP_E_INVALID, "null blocks");
    if (blocks_dtype != TAP_DT_F32 && blocks_dtype != TAP_DT_I32)
        return tap_fail(ctx, TAP_E_INVALID, "bad blocks_dtype %d", blocks_dtype);
    StepArgs a = {};
    a.blocks = blocks; a.blocks_dtype = blocks_dtype; a.active = active; a.feature_out = feature_out;
    return step_common(ctx, d, state, a, stream);
}

extern "C" int tap_env_step_gather(tap_ctx *ctx, const tap_env_desc *d, void *state,
                                   const float *static_, int static_rows, int nR,
                                   const int64_t *ptr, const uint8_t *active, float *feature_out,
                                   void *stream)
{
    if (d && d->B == 0) return tap_desc_validate(ctx, d); // an empty batch has no buffers to check
    if (!static_ || !ptr || !d || static_rows < 1 + d->D || nR < 1)
        return tap_fail(ctx, TAP_E_INVALID, "bad gather arguments");
    StepArgs a = {};
    a.static_ = static_; a.static_rows = static_rows; a.nR = nR; a.ptr = ptr;
    a.active = active; a.feature_out = feature_out;
    return step_common(ctx, d, state, a, stream);
}

// =============================================================================================
// get_heightmap / calc_ratio / attribute export / error check
// =============================================================================================

template <int D, int G>
__global__ void __launch_bounds__(TAP_BLOCK) k_env_feature(tap_env_desc d, EnvView v, float *out, int flen)
{
    __shared__ int s[TAP_BLOCK];
    const int tid = threadIdx.x, grp = tid / G, cell = tid % G;
    const int env = blockIdx.x * (TAP_BLOCK / G) + grp;
    const int cells = d.W * d.L;
    const bool ev = env < d.B;
    const int hm = (ev && cell < cells) ? v.hm[(size_t)env * cells + cell] : 0;
    s[tid] = hm;
    __syncthreads();
    // out-of-range groups run the writer on a dummy row so cross-lane ops stay convergent
    float dummy;
    (void)dummy;
    if (ev) tap_write_feature<D, G>(d.feature, d.W, d.L, s + grp * G, cell, hm, out + (size_t)env * flen);
    else if (d.feature == TAP_FEAT_ZERO) (void)group_min<G>(INT_MAX);
}

template <int D, int G>
static int launch_feature(tap_ctx *ctx, const tap_env_desc *d, const EnvView &v, float *out, hipStream_t st)
{
    const int epb = TAP_BLOCK / G, grid = (d->B + epb - 1) / epb;
    if (grid == 0) return TAP_OK;
    hipLaunchKernelGGL((k_env_feature<D, G>), dim3(grid), dim3(TAP_BLOCK), 0, st, *d, v, out,
                       tap_env_feature_len(d));
    TAP_LAUNCH_CHECK(ctx, "k_env_feature");
    return TAP_OK;
}

extern "C" int tap_env_feature(tap_ctx *ctx, const tap_env_desc *d, const void *state,
                               float *feature_out, void *stream)
{
    int rc = tap_desc_validate(ctx, d);
    if (rc) return rc;
    if (d->B == 0) return TAP_OK; // an empty batch has no buffers to check
    if (!state || !feature_out) return tap_fail(ctx, TAP_E_INVALID, "null pointer");
    EnvView v;
    tap_env_layout(d, const_cast<void *>(state), &v);
    if (tap_is_big(d) || tap_is_big_macs(d) || tap_is_big_macs3(d) || d->strategy == TAP_LB)
        return tap_big_feature(ctx, d, v, feature_out, tap_env_feature_len(d), (hipStream_t)stream);
    TAP_DISPATCH_DG(launch_feature, d, ctx, d, v, feature_out, (hipStream_t)stream);
}

// tools.py:3887-3966, one thread per env
__global__ void __launch_bounds__(TAP_BLOCK) k_env_ratio(tap_env_desc d, EnvView v, float *r32,
                                                         double *r64, double *cps)
{
    const int env = blockIdx.x * TAP_BLOCK + threadIdx.x;
    if (env >= d.B) return;
    const int cells = d.W * d.L;
    const int4 c = reinterpret_cast<const int4 *>(v.cnt)[env];
    double C = 0.0, P = 0.0, S = 0.0;
    if (c.w != 0) {                                               // :3888-3889
        int height = 0;
        for (int i = 0; i < cells; ++i) height = max(height, v.hm[(size_t)env * cells + i]);
        const long long box = (long long)d.W * d.L * height;      // :3893-3896
        C = (double)c.x / (double)box;                            // :3902
        P = (double)c.x / (double)(c.y + c.x);                    // :3903
        S = (double)c.z / (double)c.w;                            // :3904
    }
    const double r = tap_ratio_formula(d.ratio_mode, C, P, S);
    if (r32) r32[env] = (float)r;                                 // model.py:499,510 fp32 store
    if (r64) r64[env] = r;
    if (cps) { cps[(size_t)env * 3] = C; cps[(size_t)env * 3 + 1] = P; cps[(size_t)env * 3 + 2] = S; }
}

extern "C" int tap_env_ratio(tap_ctx *ctx, const tap_env_desc *d, const void *state,
                             float *ratio_out, double *ratio64_out, double *cps_out, void *stream)
{
    int rc = tap_desc_validate(ctx, d);
    if (rc) return rc;
    if (d->B == 0) return TAP_OK; // an empty batch has no buffers to check
    if (!state) return tap_fail(ctx, TAP_E_INVALID, "null state");
    EnvView v;
    tap_env_layout(d, const_cast<void *>(state), &v);
    const int grid = (d->B + TAP_BLOCK - 1) / TAP_BLOCK;
    if (grid == 0) return TAP_OK;
    hipLaunchKernelGGL(k_env_ratio, dim3(grid), dim3(TAP_BLOCK), 0, (hipStream_t)stream, *d, v,
                       ratio_out, ratio64_out, cps_out);
    TAP_LAUNCH_CHECK(ctx, "k_env_ratio");
    return TAP_OK;
}

__global__ void __launch_bounds__(TAP_BLOCK) k_env_export(tap_env_desc d, EnvView v, int32_t *hm,
                                                          int32_t *pos, uint8_t *st, int32_t *cnt)
{
    const int env = blockIdx.x * TAP_BLOCK + threadIdx.x;
    if (env >= d.B) return;
    const int cells = d.W * d.L, D = d.D, n = d.n_max;
    const size_t B = (size_t)d.B;
    if (hm) for (int i = 0; i < cells; ++i) hm[(size_t)env * cells + i] = v.hm[(size_t)env * cells + i];
    const int done = v.cnt[(size_t)env * 4 + 3]; // steps taken; later rows may hold a previous episode
    if (pos) for (int t = 0; t < n * D; ++t) pos[(size_t)env * n * D + t] = (t / D < done) ? v.pos[(size_t)t * B + env] : 0;
    if (st) for (int t = 0; t < n; ++t) st[(size_t)env * n + t] = (t < done) ? v.stable[(size_t)t * B + env] : 0;
    if (cnt) for (int k = 0; k < 4; ++k) cnt[(size_t)env * 4 + k] = v.cnt[(size_t)env * 4 + k];
}

extern "C" int tap_env_export(tap_ctx *ctx, const tap_env_desc *d, const void *state,
                              int32_t *heightmap_out, int32_t *positions_out, uint8_t *stable_out,
                              int32_t *counters_out, void *stream)
{
    int rc = tap_desc_validate(ctx, d);
    if (rc) return rc;
    if (d->B == 0) return TAP_OK; // an empty batch has no buffers to check
    if (!state) return tap_fail(ctx, TAP_E_INVALID, "null state");
    EnvView v;
    tap_env_layout(d, const_cast<void *>(state), &v);
    const int grid = (d->B + TAP_BLOCK - 1) / TAP_BLOCK;
    if (grid == 0) return TAP_OK;
    hipLaunchKernelGGL(k_env_export, dim3(grid), dim3(TAP_BLOCK), 0, (hipStream_t)stream, *d, v,
                       heightmap_out, positions_out, stable_out, counters_out);
    TAP_LAUNCH_CHECK(ctx, "k_env_export");
    return TAP_OK;
}

__global__ void __launch_bounds__(TAP_BLOCK) k_env_errors(int B, const int32_t *err, int32_t *out)
{
    const int env = blockIdx.x * TAP_BLOCK + threadIdx.x;
    if (env < B) out[env] = err[env];
}

extern "C" int tap_env_errors(tap_ctx *ctx, const tap_env_desc *d, const void *state, int32_t *err_out, void *stream)
{
    int rc = tap_desc_validate(ctx, d);
    if (rc) return rc;
    if (d->B == 0) return TAP_OK; // an empty batch has no buffers to check
    if (!state || !err_out) return tap_fail(ctx, TAP_E_INVALID, "null pointer");
    EnvView v;
    tap_env_layout(d, const_cast<void *>(state), &v);
    hipLaunchKernelGGL(k_env_errors, dim3((d->B + TAP_BLOCK - 1) / TAP_BLOCK), dim3(TAP_BLOCK), 0, (hipStream_t)stream, d->B, v.err, err_out);
    TAP_LAUNCH_CHECK(ctx, "k_env_errors");
    return TAP_OK;
}

__global__ void __launch_bounds__(TAP_BLOCK) k_env_check(int B, const int32_t *err, int32_t *out)
{
    int bad = 0, bits = 0;
    for (int i = blockIdx.x * TAP_BLOCK + threadIdx.x; i < B; i += gridDim.x * TAP_BLOCK) {
        const int e = err[i];
        bad += e != 0;
        bits |= e;
    }
    for (int o = 32; o > 0; o >>= 1) { bad += __shfl_xor(bad, o); bits |= __shfl_xor(bits, o); }
    if ((threadIdx.x & 63) == 0 && bad) { atomicAdd(&out[0], bad); atomicOr(&out[1], bits); }
}

extern "C" int tap_env_check(tap_ctx *ctx, const tap_env_desc *d, const void *state,
                             int32_t *n_bad_out, void *stream)
{
    if (n_bad_out) *n_bad_out = 0;
    int rc = tap_desc_validate(ctx, d);
    if (rc) return rc;
    if (d->B == 0) return TAP_OK; // an empty batch has no buffers to check
    if (!state) return tap_fail(ctx, TAP_E_INVALID, "null state");
    EnvView v;
    tap_env_layout(d, const_cast<void *>(state), &v);
    // the sticky error words are reduced on the device (count of flagged envs, OR of their bits): the host
    // reads 8 bytes, not B words
    int32_t host[2] = {0, 0};
    hipStream_t st = (hipStream_t)stream;
    // one of TAP_CHK_SLOTS pairs of ints per call, handed out round-robin: checks issued back to back on different
    // streams (or from two threads, against the "a ctx is not thread-safe" rule) do not reduce into each other
    int32_t *chk = ctx->chk + 2 * (__atomic_fetch_add(&ctx->chk_next, 1u, __ATOMIC_RELAXED) % TAP_CHK_SLOTS);
    hipError_t e = hipMemsetAsync(chk, 0, 2 * sizeof(int32_t), st);
    if (e == hipSuccess) {
        const int grid = min((d->B + TAP_BLOCK - 1) / TAP_BLOCK, 1024);
        hipLaunchKernelGGL(k_env_check, dim3(grid), dim3(TAP_BLOCK), 0, st, d->B, v.err, chk);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, chk, sizeof(host), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) return tap_fail(ctx, TAP_E_HIP, "error-word readback failed: %s", hipGetErrorString(e));
    const int bad = host[0], bits = host[1];
    if (n_bad_out) *n_bad_out = bad;
    if (bits & 1) return tap_fail(ctx, TAP_E_OVERFLOW, "%d container(s) exceeded height H=%d", bad, d->H);
    if (bits & 2) return tap_fail(ctx, TAP_E_STEPS, "%d container(s) stepped more than blocks_num=%d times", bad, d->n_max);
    if (bits & 4) return tap_fail(ctx, TAP_E_INVALID, "%d container(s) were given a block side < 1 or a column index outside [0, nR)", bad);
    if (bits & 8) return tap_fail(ctx, TAP_E_OVERFLOW, "%d container(s) reached a state in which the reference's MACS code raises (tools.py:2550 IndexError / :2865 UnboundLocalError)", bad);
    if (bits & 16) return tap_fail(ctx, TAP_E_UNSUPPORTED, "%d container(s) exceeded the MACS candidate-list capacity", bad);
    return TAP_OK;
}
