// tap_masks.h -- device helpers shared by masks.hip and transition.hip (pack.py:276-376).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

struct MaskArgs {
    int B, n, R, nR, rows, update_rows, static_rows;
    const float *dyn_in;
    float *dyn_out;
    const float *static_;
    const int64_t *ptr;
    const float *mask_in;
    const float *cs_in;
    float *cs_out;
    float *cur_out;
    float *mask_out;
};

// mask math for one column: pack.py:318-329
__device__ __forceinline__ void mask_column(const MaskArgs &a, int env, int j, long real_m,
                                            float move, float small, float large)
{
    float keep = a.mask_in ? a.mask_in[(size_t)env * a.nR + j] : 1.f;
    if (a.ptr)
        for (int r = 0; r < a.R; ++r)
            if (j == real_m + (long)a.n * r) keep = 0.f;            // :320-321
    if (a.mask_out) a.mask_out[(size_t)env * a.nR + j] = keep;    // chosen_mask
    const float dm = small * large + move;                        // :327-328
    if (a.cur_out) a.cur_out[(size_t)env * a.nR + j] = dm != 0.f ? 0.f : keep; // :329
}


// flat float ranges [lo_i, hi_i) of the rows update_dynamic clears for one env (pack.py:372-374)
struct ClearRanges {
    int lo[3], hi[3]; // a slab has < 2^31 floats
};

__device__ __forceinline__ ClearRanges clear_ranges(const MaskArgs &a, long real)
{
    ClearRanges c;
    for (int i = 0; i < 3; ++i) {
        const long r = real + (long)a.n * i;
        const bool on = i < a.update_rows && real >= 0 && r < a.rows;
        c.lo[i] = on ? (int)(r * a.nR) : -1;
        c.hi[i] = on ? (int)((r + 1) * a.nR) : -1;
    }
    return c;
}

__device__ __forceinline__ bool in_cleared(const ClearRanges &c, int f)
{
    return (f >= c.lo[0] && f < c.hi[0]) || (f >= c.lo[1] && f < c.hi[1]) || (f >= c.lo[2] && f < c.hi[2]);
}

// incremental column sums + both masks for one env, lanes j = lane, lane+64, ... (pack.py:318-329)
__device__ __forceinline__ void mask_env(const MaskArgs &a, int env, int lane, long real, long p)
{
    const int nR = a.nR;
    const size_t slab = (size_t)a.rows * nR;
    long real_m = p;
    while (real_m >= a.n) real_m -= a.n;                          // pack.py:314-316
    for (int j = lane; j < nR; j += 64) {
        float sum[3], row[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {                             // issue all six loads first
            sum[s] = a.cs_in[((size_t)env * 3 + s) * nR + j];
            const long r = real + (long)a.n * s;
            row[s] = (a.dyn_out && s < a.update_rows && real >= 0 && r < a.rows)
                         ? a.dyn_in[(size_t)env * slab + (size_t)r * nR + j] : 0.f; // the row being cleared
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            sum[s] -= row[s];
            if (a.cs_out) a.cs_out[((size_t)env * 3 + s) * nR + j] = sum[s];
        }
        mask_column(a, env, j, real_m, sum[0], sum[1], sum[2]);
    }
}

// ---- fast path of one "stream wave" ----------------------------------------------------------
// Copies NS consecutive env slabs (contiguous in memory) out of place with the chosen rows cleared
// (pack.py:370-374), updates the column-sum shadow and writes both masks (pack.py:318-329), with a
// SINGLE memory round trip before the stores:
//   * every small input (ptr, row 0 of static, old column sums, old mask) is loaded up front, next
//     to the first batch of slab loads; the block id `real = static[b,0,ptr]` (pack.py:339) is
//     picked out of the row-0 registers with a wave shuffle instead of a dependent load;
//   * slab loads are unconditional (clamped index) and issued U per lane before any store, so the
//     compiler can count them (no vmcnt(0) stalls) and a wave has U KiB in flight;
//   * the rows being cleared are captured into a wave-private LDS tile on their way through the
//     registers, so "new sum = old sum - cleared row" needs no second read of the slab.
// Requirements (checked by the caller): nR % 4 == 0, 16-byte aligned tensors, nR <= 64 * NC.
// lds: NS * 3 * nR floats private to this wave.  NC = columns per lane (1, 2 or 4).
template <int NS, int U, int NC>
__device__ __forceinline__ void stream_wave_fast(const MaskArgs &a, int senv0, int lane,
                                                 const bool (&on)[NS], float *lds)
{
    const int nR = a.nR;
    const size_t slab = (size_t)a.rows * nR;
    const int nchunk = (int)(slab / 4);
    int total = 0;
#pragma unroll
    for (int k = 0; k < NS; ++k) if (on[k]) total = (k + 1) * nchunk; // on[] is a prefix
    if (total == 0) return;
    long p[NS];
    float row0[NS][NC], cs[NS][NC][3], keep[NS][NC];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int env = senv0 + k;
        p[k] = on[k] ? (long)a.ptr[env] : 0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = lane + 64 * c;
            const bool ok = on[k] && j < nR;
            row0[k][c] = ok ? a.static_[(size_t)env * a.static_rows * nR + j] : 0.f;
            keep[k][c] = ok ? (a.mask_in ? a.mask_in[(size_t)env * nR + j] : 1.f) : 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) cs[k][c][s] = ok ? a.cs_in[((size_t)env * 3 + s) * nR + j] : 0.f;
        }
    }
    for (int i = lane; i < NS * 3 * nR; i += 64) lds[i] = 0.f;

    const float4 *s4 = reinterpret_cast<const float4 *>(a.dyn_in + (size_t)senv0 * slab);
    float4 *d4 = reinterpret_cast<float4 *>(a.dyn_out + (size_t)senv0 * slab);
    ClearRanges cr[NS];
    bool have_real = false;
    for (int base = 0; base < total; base += 64 * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = s4[min(base + u * 64 + lane, total - 1)];
        if (!have_real) {
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                float r0 = 0.f;                                               // pack.py:339 via shuffle
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float t = __shfl(row0[k][c], (int)(p[k] & 63));
                    if ((p[k] >> 6) == c) r0 = t;
                }
                cr[k] = clear_ranges(a, on[k] ? (long)r0 : -1);
            }
            have_real = true;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = base + u * 64 + lane;
            if (q < total) {
                const int k = (NS > 1 && q >= nchunk) ? 1 : 0;                // NS <= 2
                const int f = (q - k * nchunk) * 4;
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    if (f >= cr[k].lo[s] && f < cr[k].hi[s]) {
                        *reinterpret_cast<float4 *>(lds + (k * 3 + s) * nR + (f - cr[k].lo[s])) = v[u];
                        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                d4[q] = v[u];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        if (!on[k]) continue;
        const int env = senv0 + k;
        long real_m = p[k];
        while (real_m >= a.n) real_m -= a.n;                                  // pack.py:314-316
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = lane + 64 * c;
            if (j >= nR) continue;
            float sum[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                sum[s] = cs[k][c][s] - lds[(k * 3 + s) * nR + j];
                if (a.cs_out) a.cs_out[((size_t)env * 3 + s) * nR + j] = sum[s];
            }
            float kp = keep[k][c];
            for (int r = 0; r < a.R; ++r)
                if (j == real_m + (long)a.n * r) kp = 0.f;                    // pack.py:320-321
            if (a.mask_out) a.mask_out[(size_t)env * nR + j] = kp;
            const float dm = sum[1] * sum[2] + sum[0];                        // pack.py:327-328
            if (a.cur_out) a.cur_out[(size_t)env * nR + j] = dm != 0.f ? 0.f : kp; // pack.py:329
        }
    }
}

// columns per lane the fast path needs: 1, 2 or 4; 0 = use the generic element-wise path
inline int mask_fast_path_cols(const MaskArgs &a)
{
    const bool ok = a.dyn_out && a.ptr && a.static_ && a.cs_in && (a.nR % 4 == 0) && a.nR <= 256 && a.rows >= 1 &&
                    ((reinterpret_cast<uintptr_t>(a.dyn_in) | reinterpret_cast<uintptr_t>(a.dyn_out)) % 16 == 0);
    return !ok ? 0 : a.nR <= 64 ? 1 : a.nR <= 128 ? 2 : 4;
}
