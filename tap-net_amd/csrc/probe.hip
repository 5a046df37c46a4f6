// probe.hip -- bandwidth calibration kernels for the roofline figures bench.py prints ("confirm the
// peak on the box with a device copy kernel").  Not part of the hot path: scripts/calibrate_bw.py times them and
// commits the result under profiles/; bench.py reads that file for the measured read+write and write-only ceilings.
// The access shapes are the hot kernels' own: 16 bytes per lane, a wave covers 1 KiB contiguous, plain or
// nontemporal stores.  gfx950 only.
#include "tap_common.h"

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int PROBE_UNROLL = 4; // float4 per thread

// kind 0: copy (plain stores)  1: copy (nontemporal stores)  2: fill (plain)  3: fill (nontemporal)  4: read only
template <int KIND>
__global__ void __launch_bounds__(TAP_BLOCK) k_bw_probe(v4f *__restrict__ dst, const v4f *__restrict__ src, size_t n4,
                                                        float *sink)
{
    const size_t base = ((size_t)blockIdx.x * PROBE_UNROLL) * TAP_BLOCK + threadIdx.x;
    v4f v[PROBE_UNROLL];
    if (KIND == 0 || KIND == 1 || KIND == 4) {
#pragma unroll
        for (int u = 0; u < PROBE_UNROLL; ++u) {
            const size_t i = base + (size_t)u * TAP_BLOCK;
            v[u] = i < n4 ? src[i] : v4f{0.f, 0.f, 0.f, 0.f};
        }
    } else {
        const float f = (float)(threadIdx.x & 1);
#pragma unroll
        for (int u = 0; u < PROBE_UNROLL; ++u) v[u] = v4f{f, 0.f, f, 1.f};
    }
    if (KIND == 4) {
        float acc = 0.f;
#pragma unroll
        for (int u = 0; u < PROBE_UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        if (acc == 123456.75f) *sink = acc; // never true for the calibration data; keeps the loads alive
        return;
    }
#pragma unroll
    for (int u = 0; u < PROBE_UNROLL; ++u) {
        const size_t i = base + (size_t)u * TAP_BLOCK;
        if (i < n4) {
            if (KIND == 1 || KIND == 3) __builtin_nontemporal_store(v[u], dst + i);
            else dst[i] = v[u];
        }
    }
}

extern "C" int tap_bw_probe(tap_ctx *ctx, int kind, void *dst, const void *src, size_t bytes, void *stream)
{
    if (!ctx) return TAP_E_INVALID;
    if (kind < 0 || kind > 4 || bytes % 16 || ((uintptr_t)dst | (uintptr_t)src) % 16)
        return tap_fail(ctx, TAP_E_INVALID, "bw_probe: kind 0..4, 16-byte aligned buffers and sizes");
    if ((kind != 4 && !dst) || ((kind == 0 || kind == 1 || kind == 4) && !src))
        return tap_fail(ctx, TAP_E_INVALID, "bw_probe: null buffer");
    const size_t n4 = bytes / 16;
    if (n4 == 0) return TAP_OK;
    const size_t per_wg = (size_t)TAP_BLOCK * PROBE_UNROLL;
    const dim3 grid((unsigned)((n4 + per_wg - 1) / per_wg));
    hipStream_t st = (hipStream_t)stream;
    v4f *d = static_cast<v4f *>(dst);
    const v4f *s = static_cast<const v4f *>(src);
    float *sink = reinterpret_cast<float *>(ctx->chk);
    switch (kind) {
    case 0: hipLaunchKernelGGL(k_bw_probe<0>, grid, dim3(TAP_BLOCK), 0, st, d, s, n4, sink); break;
    case 1: hipLaunchKernelGGL(k_bw_probe<1>, grid, dim3(TAP_BLOCK), 0, st, d, s, n4, sink); break;
    case 2: hipLaunchKernelGGL(k_bw_probe<2>, grid, dim3(TAP_BLOCK), 0, st, d, s, n4, sink); break;
    case 3: hipLaunchKernelGGL(k_bw_probe<3>, grid, dim3(TAP_BLOCK), 0, st, d, s, n4, sink); break;
    default: hipLaunchKernelGGL(k_bw_probe<4>, grid, dim3(TAP_BLOCK), 0, st, d, s, n4, sink); break;
    }
    TAP_LAUNCH_CHECK(ctx, "k_bw_probe");
    return TAP_OK;
}
