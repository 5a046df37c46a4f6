// masks.hip -- the precedence tensors: pack.update_dynamic (pack.py:333-376), pack.update_mask
// (pack.py:276-331) and the initial mask (model.py:297-307).  gfx950 only.
//
// `dynamic` is (B, rows, nR) fp32.  One wavefront owns one env's (rows x nR) slab and streams it
// with 16-byte loads/stores (1 KiB per wave instruction, fully coalesced): the step is an
// out-of-place copy with three rows zeroed, so it is HBM-bound (2 * rows*nR*4 bytes per env).
// The three per-section column sums update_mask needs are kept in a (B, 3, nR) shadow that is
// updated incrementally (sum_new = sum_old - zeroed row), so the slab is never reduced again
// after tap_dyn_colsum built the shadow once.
#include "tap_common.h"
#include "tap_masks.h"

constexpr int WAVE = 64;
constexpr int ENVS_PER_BLOCK = TAP_BLOCK / WAVE;

// ---- full reduction: colsum[b, s, j] = sum_i dynamic[b, s*n + i, j] -------------------------
__global__ void __launch_bounds__(TAP_BLOCK) k_dyn_colsum(int B, int n, int nR, int rows,
                                                          const float *__restrict__ dyn,
                                                          float *__restrict__ cs)
{
    const int env = blockIdx.x * ENVS_PER_BLOCK + threadIdx.x / WAVE;
    const int lane = threadIdx.x % WAVE;
    if (env >= B) return;
    const float *slab = dyn + (size_t)env * rows * nR;
    for (int j = lane; j < nR; j += WAVE) {
        for (int s = 0; s < 3; ++s) {
            float acc = 0.f;
            for (int i = s * n; i < (s + 1) * n && i < rows; ++i) acc += slab[(size_t)i * nR + j];
            cs[((size_t)env * 3 + s) * nR + j] = acc;
        }
    }
}

// ---- bit shadow: bits[b, j] bit r = dynamic[b, r, j] != 0; counts the elements that are not 0 / 1 ----
__global__ void __launch_bounds__(TAP_BLOCK) k_dyn_bits(int B, int nR, int rows, const float *__restrict__ dyn,
                                                        unsigned long long *__restrict__ bits, int *nonbinary)
{
    const int env = blockIdx.x * ENVS_PER_BLOCK + threadIdx.x / WAVE;
    const int lane = threadIdx.x % WAVE;
    if (env >= B) return;
    const float *slab = dyn + (size_t)env * rows * nR;
    int bad = 0;
    for (int j = lane; j < nR; j += WAVE) {
        unsigned long long w = 0, w2 = 0;
        for (int r = 0; r < rows; ++r) {
            const float v = slab[(size_t)r * nR + j];
            if (r < 64) w |= (unsigned long long)(v != 0.f) << r;
            else w2 |= (unsigned long long)(v != 0.f) << (r & 63);
            bad += (v != 0.f && v != 1.f);
        }
        if (bits) {                                        // null: count only (any number of rows)
            if (rows <= 64) bits[(size_t)env * nR + j] = w;
            else { bits[(size_t)env * 2 * nR + j] = w; bits[((size_t)env * 2 + 1) * nR + j] = w2; } // two planes
        }
    }
    if (bad && nonbinary) atomicAdd(nonbinary, bad);
}

// ---- fused step: copy-with-zeroed-rows + incremental column sums + both masks ----------------
// NC > 0 = stream_wave_fast with NC columns per lane (nR % 4 == 0, nR <= 256, aligned, shadow
// given); NC == 0 = the generic
// element-wise path, which also serves tap_update_mask (no copy) and tap_update_dynamic without a
// shadow.
// MODE: 0 = fp32 copy with the column-sum shadow (or the generic path), 1 = bit shadow, 2 = first step;
// 3 / 4 = the same two on the two-word shadow (65 .. 128 rows)
template <int NC, int MODE>
__global__ void __launch_bounds__(TAP_BLOCK) k_mask_step(TAP_MASK_HOT_PARAMS, MaskArgs ka)
{
    extern __shared__ float mask_lds[];
    const int wave = threadIdx.x / WAVE;
    const int env = blockIdx.x * ENVS_PER_BLOCK + wave;
    const int lane = threadIdx.x % WAVE;
    if (env >= h_B) return;
    const MaskArgs a = tap_mask_hot(ka, MODE == 1 || MODE == 3, TAP_MASK_HOT_NAMES);
    if (NC > 0) {
        const bool on[1] = {true};
        if (MODE == 3) stream_wave_bits2<(NC > 0 ? NC : 1), false>(a, env, lane, nullptr);
        else if (MODE == 4) stream_wave_bits2<(NC > 0 ? NC : 1), true>(a, env, lane, mask_lds + (size_t)wave * 4 * a.nR);
        else if (MODE == 1) stream_wave_bits<1, (NC > 0 ? NC : 1), false>(a, env, lane, on, mask_lds + (size_t)wave * 3 * a.nR);
        else if (MODE == 2) stream_wave_bits<1, (NC > 0 ? NC : 1), true>(a, env, lane, on, mask_lds + (size_t)wave * 3 * a.nR);
        else stream_wave_fast<1, (NC > 2 ? 4 : 6), (NC > 0 ? NC : 1)>(a, env, lane, on, mask_lds + (size_t)wave * 3 * a.nR);
        return;
    }
    const int nR = a.nR;
    const size_t slab = (size_t)a.rows * nR;
    bool badp = false;
    const long pc = a.ptr ? tap_col((long)a.ptr[env], nR, badp) : 0;
    // pack.py:339: block id read from row 0 of `static` as float -> long; an index outside [0, nR) (the
    // reference's gather raises) clears nothing and removes no column
    const long real = (a.ptr && a.static_ && !badp) ? (long)a.static_[(size_t)env * a.static_rows * nR + pc] : -1;
    const long p = (a.ptr && !badp) ? pc : -1;
    if (a.dyn_out) {
        const ClearRanges cr = clear_ranges(a, real);
        const float *src = a.dyn_in + (size_t)env * slab;
        float *dst = a.dyn_out + (size_t)env * slab;
        for (long f = lane; f < (long)slab; f += WAVE) {
            float v = src[f];
            if (in_cleared(cr, (int)f)) v = 0.f;
            dst[f] = v;
        }
    }
    if (a.cs_out || a.cur_out || a.mask_out) mask_env(a, env, lane, real, p);
}

static int launch_mask_step(tap_ctx *ctx, const MaskArgs &a, hipStream_t st)
{
    const int grid = (a.B + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK;
    if (grid == 0) return TAP_OK;
    const bool wide = (a.bits_in || mask_builds_bits(a)) && a.rows > 64;     // two words per column
    const size_t lds = (size_t)ENVS_PER_BLOCK * (wide ? 4 : 3) * a.nR * sizeof(float);
    const int mode = (a.bits_in ? 1 : mask_builds_bits(a) ? 2 : 0) + (wide ? 2 : 0);
#define TAP_LAUNCH_T(NC_, M_, LDS_) hipLaunchKernelGGL((k_mask_step<NC_, M_>), dim3(grid), dim3(TAP_BLOCK), LDS_, st, TAP_MASK_HOT_ARGS(a), a)
#define TAP_LAUNCH_M(NC_) do { if (mode == 1) TAP_LAUNCH_T(NC_, 1, lds); else if (mode == 2) TAP_LAUNCH_T(NC_, 2, lds); else if (mode == 3) TAP_LAUNCH_T(NC_, 3, lds); else if (mode == 4) TAP_LAUNCH_T(NC_, 4, lds); else TAP_LAUNCH_T(NC_, 0, lds); } while (0)
    switch (mask_fast_path_cols(a)) {
    case 1: TAP_LAUNCH_M(1); break;
    case 2: TAP_LAUNCH_M(2); break;
    case 4: TAP_LAUNCH_M(4); break;
    default: TAP_LAUNCH_T(0, 0, 0); break;
    }
#undef TAP_LAUNCH_M
#undef TAP_LAUNCH_T
    TAP_LAUNCH_CHECK(ctx, "k_mask_step");
    return TAP_OK;
}

static int check_shape(tap_ctx *ctx, int B, int n, int nR, int rows)
{
    if (B < 0 || n < 1 || nR < 1 || rows < 1 || nR % n != 0)
        return tap_fail(ctx, TAP_E_INVALID, "bad mask shape B=%d n=%d nR=%d rows=%d", B, n, nR, rows);
    return TAP_OK;
}

extern "C" int tap_dyn_colsum(tap_ctx *ctx, int B, int n, int nR, int rows, const float *dynamic,
                              float *colsum_out, void *stream)
{
    int rc = check_shape(ctx, B, n, nR, rows);
    if (rc) return rc;
    if (B == 0) return TAP_OK;
    if (!dynamic || !colsum_out) return tap_fail(ctx, TAP_E_INVALID, "null pointer");
    const int grid = (B + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK;
    if (grid == 0) return TAP_OK;
    hipLaunchKernelGGL(k_dyn_colsum, dim3(grid), dim3(TAP_BLOCK), 0, (hipStream_t)stream, B, n, nR,
                       rows, dynamic, colsum_out);
    TAP_LAUNCH_CHECK(ctx, "k_dyn_colsum");
    return TAP_OK;
}

extern "C" int tap_bits_words(int rows, int nR)
{
    if (rows < 1 || rows > 128 || nR < 1 || nR > 256 || nR % 4 != 0) return 0;
    return mask_bit_planes(rows) * nR;
}

extern "C" int tap_dyn_bits(tap_ctx *ctx, int B, int nR, int rows, const float *dynamic,
                            unsigned long long *bits_out, int32_t *nonbinary_out, void *stream)
{
    if (B < 0 || nR < 1 || rows < 1) return tap_fail(ctx, TAP_E_INVALID, "bad shape B=%d nR=%d rows=%d", B, nR, rows);
    if (B == 0) return TAP_OK;
    if (rows > 128 && bits_out) return tap_fail(ctx, TAP_E_UNSUPPORTED, "the bit shadow holds at most 128 rows (rows=%d)", rows);
    if (!dynamic || (!bits_out && !nonbinary_out)) return tap_fail(ctx, TAP_E_INVALID, "null pointer");
    const int grid = (B + ENVS_PER_BLOCK - 1) / ENVS_PER_BLOCK;
    if (grid == 0) return TAP_OK;
    hipLaunchKernelGGL(k_dyn_bits, dim3(grid), dim3(TAP_BLOCK), 0, (hipStream_t)stream, B, nR, rows, dynamic,
                       bits_out, nonbinary_out);
    TAP_LAUNCH_CHECK(ctx, "k_dyn_bits");
    return TAP_OK;
}

extern "C" int tap_mask_step_bits(tap_ctx *ctx, int B, int n, int R, int rows, int update_rows,
                                  const unsigned long long *bits_in, const float *static_, int static_rows,
                                  const int64_t *ptr, const float *mask_in, unsigned long long *bits_out,
                                  float *dyn_out, float *current_out, float *mask_out, void *stream)
{
    int rc = check_shape(ctx, B, n, n * R, rows);
    if (rc) return rc;
    if (B == 0) return TAP_OK;
    if (!bits_in || (ptr && (!static_ || static_rows < 1)) || update_rows < 0 || update_rows > 3 ||
        (!ptr && update_rows != 0) || bits_in == bits_out || (!bits_out && !dyn_out && !current_out && !mask_out))
        return tap_fail(ctx, TAP_E_INVALID, "bad mask_step_bits arguments");
    MaskArgs a = mask_finish(MaskArgs{B, n, R, n * R, rows, update_rows, static_rows, nullptr, dyn_out, static_, ptr,
                  mask_in, nullptr, nullptr, current_out, mask_out, bits_in, bits_out});
    if (!mask_bits_ok(a))
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "bit shadow needs nR %% 4 == 0, nR <= 256, rows <= 128, 16-byte aligned buffers");
    return launch_mask_step(ctx, a, (hipStream_t)stream);
}

extern "C" int tap_mask_step_first(tap_ctx *ctx, int B, int n, int R, int rows, int update_rows,
                                   const float *dyn_in, const float *static_, int static_rows, const int64_t *ptr,
                                   const float *mask_in, unsigned long long *bits_out, float *dyn_out,
                                   float *current_out, float *mask_out, int32_t *nonbinary_out, void *stream)
{
    int rc = check_shape(ctx, B, n, n * R, rows);
    if (rc) return rc;
    if (B == 0) return TAP_OK;
    if (!dyn_in || !bits_out || (ptr && (!static_ || static_rows < 1)) || update_rows < 0 || update_rows > 3 ||
        (!ptr && update_rows != 0) || dyn_in == dyn_out)
        return tap_fail(ctx, TAP_E_INVALID, "bad mask_step_first arguments");
    MaskArgs a = mask_finish(MaskArgs{B, n, R, n * R, rows, update_rows, static_rows, dyn_in, dyn_out, static_, ptr,
                  mask_in, nullptr, nullptr, current_out, mask_out, nullptr, bits_out, nonbinary_out});
    if (!mask_bits_ok(a))
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "bit shadow needs nR %% 4 == 0, nR <= 256, rows <= 128, 16-byte aligned buffers");
    return launch_mask_step(ctx, a, (hipStream_t)stream);
}

extern "C" int tap_update_dynamic(tap_ctx *ctx, int B, int n, int nR, int rows, int update_rows,
                                  const float *dyn_in, const float *static_, int static_rows,
                                  const int64_t *ptr, float *dyn_out, const float *colsum_in,
                                  float *colsum_out, void *stream)
{
    int rc = check_shape(ctx, B, n, nR, rows);
    if (rc) return rc;
    if (B == 0) return TAP_OK;
    if (!dyn_in || !static_ || !ptr || !dyn_out || static_rows < 1 || update_rows < 0 || update_rows > 3)
        return tap_fail(ctx, TAP_E_INVALID, "bad update_dynamic arguments");
    if (dyn_in == dyn_out) return tap_fail(ctx, TAP_E_INVALID, "update_dynamic is out of place (pack.py:370)");
    if ((colsum_in == nullptr) != (colsum_out == nullptr))
        return tap_fail(ctx, TAP_E_INVALID, "colsum_in and colsum_out go together");
    MaskArgs a = mask_finish(MaskArgs{B, n, nR / n, nR, rows, update_rows, static_rows, dyn_in, dyn_out, static_, ptr,
                  nullptr, colsum_in, colsum_out, nullptr, nullptr});
    return launch_mask_step(ctx, a, (hipStream_t)stream);
}

extern "C" int tap_update_mask(tap_ctx *ctx, int B, int n, int R, const float *mask_in,
                               const float *colsum, const int64_t *ptr, float *current_out,
                               float *mask_out, void *stream)
{
    if (B < 0 || n < 1 || R < 1) return tap_fail(ctx, TAP_E_INVALID, "bad mask shape");
    if (B == 0) return TAP_OK;
    if (!colsum || (!current_out && !mask_out)) return tap_fail(ctx, TAP_E_INVALID, "null pointer");
    if (ptr && !mask_in) return tap_fail(ctx, TAP_E_INVALID, "update_mask needs mask_in");
    MaskArgs a = mask_finish(MaskArgs{B, n, R, n * R, 3 * n, 0, 0, nullptr, nullptr, nullptr, ptr, mask_in, colsum,
                  nullptr, current_out, mask_out});
    return launch_mask_step(ctx, a, (hipStream_t)stream);
}

extern "C" int tap_mask_step(tap_ctx *ctx, int B, int n, int R, int rows, int update_rows,
                             const float *dyn_in, const float *static_, int static_rows,
                             const int64_t *ptr, const float *mask_in, const float *colsum_in,
                             float *dyn_out, float *colsum_out, float *current_out,
                             float *mask_out, void *stream)
{
    int rc = check_shape(ctx, B, n, n * R, rows);
    if (rc) return rc;
    if (B == 0) return TAP_OK;
    if (!dyn_in || !static_ || !ptr || !mask_in || !colsum_in || !dyn_out || !colsum_out ||
        !current_out || !mask_out || static_rows < 1 || update_rows < 0 || update_rows > 3)
        return tap_fail(ctx, TAP_E_INVALID, "bad mask_step arguments");
    if (dyn_in == dyn_out) return tap_fail(ctx, TAP_E_INVALID, "mask_step is out of place (pack.py:370)");
    MaskArgs a = mask_finish(MaskArgs{B, n, R, n * R, rows, update_rows, static_rows, dyn_in, dyn_out, static_, ptr,
                  mask_in, colsum_in, colsum_out, current_out, mask_out});
    return launch_mask_step(ctx, a, (hipStream_t)stream);
}
