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

// kinds 5 .. 7 (round 4): the fused step's own stores, to tell what its access SHAPE costs at the BASELINE batch.
//   5  the linear fill of kind 2 with write-through (sc0 sc1) stores
//   6  the bit-shadow expansion's shape at c2 (nR = 20, 30 rows: slabs of 2 400 B, not a multiple of a 128-byte line):
//      a 4-wave workgroup owns 8 consecutive slabs, a wave two of them; lane (rsub, c4) = (lane / 5, lane % 5) of 60
//      writes the float4 of rows rsub, rsub + 12, rsub + 24 -- store instructions of 960 / 960 / 480 contiguous bytes
//   7  the same 4 800 bytes per wave written linearly: float4 i * 64 + lane, five instructions of 1 024 ... 704 bytes
//   8  (round 5) the wave's two slabs as ONE run of 60 rows: lane (rsub, c4) of 60 writes rows rsub + 12 i, i = 0 .. 4 --
//      five instructions of 960 bytes, each starting on a 64-byte granule (the round-5 stream wave's shape)
template <int KIND>
__global__ void __launch_bounds__(TAP_BLOCK) k_bw_probe_slab(v4f *__restrict__ dst, size_t n_wg)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float f = (float)(lane & 1);
    const v4f v = {f, 0.f, f, 1.f};
    v4f *base = dst + ((size_t)blockIdx.x * 8 + wave * 2) * 150;        // 150 float4 per slab
    if (KIND == 6) {
        const int rsub = lane / 5, c4 = lane - rsub * 5;
        if (rsub < 12)
            for (int k = 0; k < 2; ++k)
                for (int r = rsub; r < 30; r += 12)
                    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(base + k * 150 + r * 5 + c4), "v"(v) : "memory");
    } else if (KIND == 8) {
        if (lane < 60)
            for (int q = lane; q < 300; q += 60)
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(base + q), "v"(v) : "memory");
    } else {
        for (int q = lane; q < 300; q += 64)
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(base + q), "v"(v) : "memory");
    }
}

template <int KIND>
__global__ void __launch_bounds__(TAP_BLOCK) k_bw_probe_wt(v4f *__restrict__ dst, size_t n4)
{
    const size_t base = ((size_t)blockIdx.x * PROBE_UNROLL) * TAP_BLOCK + threadIdx.x;
    const float f = (float)(threadIdx.x & 1);
    const v4f v = {f, 0.f, f, 1.f};
#pragma unroll
    for (int u = 0; u < PROBE_UNROLL; ++u) {
        const size_t i = base + (size_t)u * TAP_BLOCK;
        if (i < n4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(dst + i), "v"(v) : "memory");
    }
}

// kinds 9 .. 13 (round 5): SPARSE reads, to calibrate what rocprofv3's FETCH_SIZE tallies for the rolling step's
// scattered records (the guide's "x 2" holds for wide coalesced reads only).  Record r of the region starts at
// r * STRIDE bytes and TOUCH of its bytes are read (8: one lane's u64; 16 k: k lanes' 16 bytes each), every record once.
//   9: 8 of 128    10: 32 of 128    11: 32 of 64    12: 64 of 128    13: 32 of 256
template <int TOUCH, int STRIDE>
__global__ void __launch_bounds__(TAP_BLOCK) k_fetch_probe(const char *__restrict__ src, size_t records, float *sink)
{
    constexpr int LPR = TOUCH >= 16 ? TOUCH / 16 : 1;                 // lanes per record
    const size_t t = (size_t)blockIdx.x * TAP_BLOCK + threadIdx.x;
    const size_t r = t / LPR;
    if (r >= records) return;
    const char *p = src + r * STRIDE + (t % LPR) * 16;
    float acc;
    if (TOUCH == 8) {
        const unsigned long long w = *reinterpret_cast<const unsigned long long *>(p);
        acc = __uint_as_float((unsigned)(w ^ (w >> 32)));
    } else {
        const v4f v = *reinterpret_cast<const v4f *>(p);
        acc = v.x + v.y + v.z + v.w;
    }
    if (acc == 123456.75f) *sink = acc;                               // never true for the calibration data
}
template <int TOUCH, int STRIDE>
static void fetch_probe_launch(const void *src, size_t bytes, float *sink, hipStream_t st)
{
    const size_t records = bytes / STRIDE, lanes = records * (TOUCH >= 16 ? TOUCH / 16 : 1);
    hipLaunchKernelGGL((k_fetch_probe<TOUCH, STRIDE>), dim3((unsigned)((lanes + TAP_BLOCK - 1) / TAP_BLOCK)), dim3(TAP_BLOCK), 0, st,
                       static_cast<const char *>(src), records, sink);
}

extern "C" int tap_bw_probe(tap_ctx *ctx, int kind, void *dst, const void *src, size_t bytes, void *stream)
{
    if (!ctx) return TAP_E_INVALID;
    if (kind >= 9 && kind <= 13) {
        if (!src || bytes % 256 || (uintptr_t)src % 256) return tap_fail(ctx, TAP_E_INVALID, "bw_probe: kinds 9 .. 13 read a 256-byte aligned region");
        if (bytes / 64 > 0x7fffffffull * 64) return tap_fail(ctx, TAP_E_INVALID, "bw_probe: region too large");
        float *snk = reinterpret_cast<float *>(ctx->chk);
        hipStream_t s_ = (hipStream_t)stream;
        if (bytes == 0) return TAP_OK;
        switch (kind) {
        case 9: fetch_probe_launch<8, 128>(src, bytes, snk, s_); break;
        case 10: fetch_probe_launch<32, 128>(src, bytes, snk, s_); break;
        case 11: fetch_probe_launch<32, 64>(src, bytes, snk, s_); break;
        case 12: fetch_probe_launch<64, 128>(src, bytes, snk, s_); break;
        default: fetch_probe_launch<32, 256>(src, bytes, snk, s_); break;
        }
        TAP_LAUNCH_CHECK(ctx, "k_fetch_probe");
        return TAP_OK;
    }
    if (kind < 0 || kind > 8 || bytes % 16 || ((uintptr_t)dst | (uintptr_t)src) % 16)
        return tap_fail(ctx, TAP_E_INVALID, "bw_probe: kind 0..8, 16-byte aligned buffers and sizes");
    if (kind >= 6 && bytes % 19200)
        return tap_fail(ctx, TAP_E_INVALID, "bw_probe: kinds 6 .. 8 write whole workgroups of 8 slabs x 2400 bytes");
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
    if (kind >= 5) {
        if (!dst) return tap_fail(ctx, TAP_E_INVALID, "bw_probe: null buffer");
        if (kind == 5) hipLaunchKernelGGL(k_bw_probe_wt<5>, grid, dim3(TAP_BLOCK), 0, st, d, n4);
        else if (kind == 6) hipLaunchKernelGGL(k_bw_probe_slab<6>, dim3((unsigned)(bytes / 19200)), dim3(TAP_BLOCK), 0, st, d, bytes / 19200);
        else if (kind == 7) hipLaunchKernelGGL(k_bw_probe_slab<7>, dim3((unsigned)(bytes / 19200)), dim3(TAP_BLOCK), 0, st, d, bytes / 19200);
        else hipLaunchKernelGGL(k_bw_probe_slab<8>, dim3((unsigned)(bytes / 19200)), dim3(TAP_BLOCK), 0, st, d, bytes / 19200);
        TAP_LAUNCH_CHECK(ctx, "k_bw_probe");
        return TAP_OK;
    }
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
