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
    long lo[3], hi[3];
};

__device__ __forceinline__ ClearRanges clear_ranges(const MaskArgs &a, long real)
{
    ClearRanges c;
    for (int i = 0; i < 3; ++i) {
        const long r = real + (long)a.n * i;
        const bool on = i < a.update_rows && real >= 0 && r < a.rows;
        c.lo[i] = on ? r * a.nR : -1;
        c.hi[i] = on ? (r + 1) * a.nR : -1;
    }
    return c;
}

__device__ __forceinline__ bool in_cleared(const ClearRanges &c, long f)
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
        float sum[3];
        for (int s = 0; s < 3; ++s) {
            float v = a.cs_in[((size_t)env * 3 + s) * nR + j];
            const long r = real + (long)a.n * s;
            if (a.dyn_out && s < a.update_rows && real >= 0 && r < a.rows)
                v -= a.dyn_in[(size_t)env * slab + (size_t)r * nR + j]; // the row being cleared
            sum[s] = v;
            if (a.cs_out) a.cs_out[((size_t)env * 3 + s) * nR + j] = v;
        }
        mask_column(a, env, j, real_m, sum[0], sum[1], sum[2]);
    }
}
