// tap_masks.h -- device helpers shared by masks.hip and transition.hip (pack.py:276-376).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>
#include <cstdlib>

constexpr size_t TAP_WT_MAX_BYTES = 64u << 20;   // write-through up to this many bytes of fp32 tensor per launch (see store_stream)

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
    // bit shadow of a 0/1-valued dynamic tensor (tap_dyn_bits): word j of env b holds column j, bit r =
    // dynamic[b, r, j] != 0.  When bits_in is set the step never reads dyn_in or the colsum shadow.
    const unsigned long long *bits_in;
    unsigned long long *bits_out;
    // bits_in null + dyn_in + bits_out set: the FIRST step on a fresh fp32 tensor -- the shadow is built from
    // dyn_in inside the step (one read of the slab); *nonbinary is incremented by the number of elements that
    // are neither 0 nor 1 (the shadow, and the step's outputs, are only valid when it stays 0)
    int *nonbinary;
    // lane mapping of the 16-byte row accesses (lane = (row group, column quad)), filled by mask_finish() on the
    // host so that no kernel divides by a run-time value: c4_magic = ceil(2^16 / (nR/4)), rp = 64 / (nR/4)
    int c4_magic, rp;
    int wt;   // flavour of the fp32 expansion's stores (store_stream), set by mask_finish()
    // 64-byte alignment of the expansion's store instructions (stream_lane_role): float4 offset of env b's slab
    // inside its 64-byte granule = (b * sb_mul + sb_add) & 3; nq = ceil(rows / rp)
    int sb_mul, sb_add, nq, nq2;   // nq2 = ceil(2 * rows / rp): a fused step's stream wave expands two slabs as one run
    // dyn_out already HOLDS the tensor the step starts from (a stepper whose two dyn phases are one buffer, tapenv.h:
    // tap_stepper_buffers): update_dynamic's result differs from its input in the chosen rows only (pack.py:370-374), so
    // the kernels that know this mode write those rows' zeros and nothing else; every other kernel ignores the hint and
    // writes the whole tensor, which is the same tensor
    int inplace;
};

inline int tap_write_through(size_t bytes);

inline MaskArgs mask_finish(MaskArgs a)
{
    const int c4 = a.nR >> 2;
    a.c4_magic = c4 > 0 ? 65536 / c4 + 1 : 0;
    a.rp = c4 > 0 ? 64 / c4 : 0;
    a.wt = tap_write_through((size_t)a.B * a.rows * a.nR * sizeof(float));
    // (rotated roles only for write-through launches: the nontemporal form keeps round 4's store loops, see stream_wave_bits)
    const bool al = a.wt && c4 > 0 && ((a.rp * c4) & 3) == 0;     // a store instruction's rp * c4 float4 are whole 64-byte granules
    a.sb_mul = al ? (a.rows * c4) & 3 : 0;
    a.sb_add = al ? (int)((reinterpret_cast<uintptr_t>(a.dyn_out) >> 4) & 3) : 0;
    a.nq = a.rp > 0 ? (a.rows + a.rp - 1) / a.rp : 0;
    a.nq2 = a.rp > 0 ? (2 * a.rows + a.rp - 1) / a.rp : 0;
    return a;
}

// ---- which rows a lane expands, and when (round 5) ------------------------------------------------------------
// The expansion's store instructions cover K = rp * (nR/4) consecutive float4 (rp whole rows) of a slab, but a
// slab of rows * nR * 4 bytes starts wherever the previous one ended: at c2 (2 400 B) three slabs in four start 32
// or 96 bytes into a 64-byte granule, at c3 / c5 (7 200 B) every other one.  Every store instruction then
// begins and ends inside a granule, and on buffers that are not cache-resident the memory side has to merge the
// pieces: a bare write-through fill in that shape takes 6.14 us per 19.66 MB against 4.02 us for stores that
// start on 64 bytes (profiles/r04_bw_store_shapes.json, kinds 6 / 7).  The cure costs no data movement between
// lanes: ROTATE the lanes' roles by sb = (slab start / 16) mod 4.  Lane l plays role (l - sb) mod K =
// (rsub, c4); the sb lanes that wrapped run one instruction late, so instruction i covers the float4
// [K*i - sb, K*i + K - sb) of the slab -- a range that starts on a granule.  Every lane still writes exactly the
// rows rsub, rsub + rp, ... of its own four columns, from words it loaded itself; only the instruction in which it
// does so moved.
struct LaneRole {
    int rsub, c4;    // role: rows rsub + rp*q of column quad c4
    int r0;          // row of the lane's store in instruction 0 (negative: the lane starts one instruction late)
    int nq;          // store instructions of the run: the caller's count, + 1 when the roles are rotated (wave-uniform)
    int sb;          // the rotation: lane l plays role (l - sb) mod K, so role (0, c4) is lane c4 + sb
};

__device__ __forceinline__ LaneRole stream_lane_role(int lane, int env, int C4, int RP, int c4_magic, int sb_mul, int sb_add, int nq)
{
    const int sb = __builtin_amdgcn_readfirstlane((env * sb_mul + sb_add) & 3);     // env is wave-uniform
    int role = lane - sb;
    const bool late = role < 0;
    role += late ? RP * C4 : 0;
    LaneRole o;
    o.rsub = (int)(((unsigned)role * (unsigned)c4_magic) >> 16);
    o.c4 = role - o.rsub * C4;
    o.r0 = o.rsub - (late ? RP : 0);
    o.nq = nq + (sb != 0);
    o.sb = sb;
    return o;
}

__host__ __device__ __forceinline__ bool mask_builds_bits(const MaskArgs &a) { return !a.bits_in && a.bits_out && a.dyn_in; }

// Kernels that run a stream wave take, AHEAD of their argument block, the fields a wave needs to ADDRESS its input loads
// (h_src = the shadow words when the step reads them, the fp32 tensor otherwise): leading scalar kernel arguments are
// preloaded into SGPRs by the dispatcher (-mllvm -amdgpu-kernarg-preload-count, gfx940+; set per file in the Makefile),
// so the loads go out without the dependent scalar-cache round trips a read of the block costs at the start of every
// wave (measured on the fused step, round 4: c2 1 214 -> 1 282 M env-steps/s, c3 490 -> 523 M).
#define TAP_MASK_HOT_PARAMS const int64_t *h_ptr, const float *h_static, const float *h_mask_in, const void *h_src, int h_B, \
                            int h_nR, int h_static_rows, int h_c4_magic
#define TAP_MASK_HOT_NAMES h_ptr, h_static, h_mask_in, h_src, h_B, h_nR, h_static_rows, h_c4_magic
// (the last word also carries sb_mul and sb_add, bits 24-25 / 26-27: the rotation of the lane roles -- tap_masks.h:
//  stream_lane_role -- enters the ADDRESS of a lane's shadow words, and read from the argument block it put one
//  scalar-cache round trip in front of every load of the wave; c4_magic <= 2^16 + 1)
#define TAP_MASK_HOT_ARGS(m) (m).ptr, (m).static_, (m).mask_in,                                                              \
        ((m).bits_in ? static_cast<const void *>((m).bits_in) : static_cast<const void *>((m).dyn_in)), (m).B, (m).nR,       \
        (m).static_rows, ((m).c4_magic | (((m).sb_mul & 3) << 24) | (((m).sb_add & 3) << 26))
// src_is_bits: compile-time in the callers (MODE == 1 / 3)
__device__ __forceinline__ MaskArgs tap_mask_hot(const MaskArgs &k, bool src_is_bits, TAP_MASK_HOT_PARAMS)
{
    MaskArgs m = k;
    m.ptr = h_ptr; m.static_ = h_static; m.mask_in = h_mask_in;
    m.B = h_B; m.nR = h_nR; m.static_rows = h_static_rows; m.c4_magic = h_c4_magic & 0xffffff;
    m.sb_mul = (h_c4_magic >> 24) & 3; m.sb_add = (h_c4_magic >> 26) & 3;
    if (src_is_bits) m.bits_in = static_cast<const unsigned long long *>(h_src);
    else m.dyn_in = static_cast<const float *>(h_src);
    return m;
}


// -DTAP_PROF (scripts/decompose_step.py): a timeline of the fused step.  Every wave of the first TAP_PROF_WGS workgroups
// records the constant-rate 100 MHz clock (s_memrealtime: comparable across CUs) at four points -- stream waves: entry,
// first store (all inputs have arrived), last store issued, stores acknowledged; placement waves: entry, state + block
// loaded, placement decided, results stored.  The product build has none of this.
#ifdef TAP_PROF
constexpr int TAP_PROF_WGS = 2048, TAP_PROF_WAVES = 16;   // 3D windows: 4 placement + 8 stream waves per workgroup (12 > 8 mixed neighbouring workgroups' stamps up to round 6)
static __device__ unsigned long long tap_prof_tl[TAP_PROF_WGS * TAP_PROF_WAVES * 4];
#define TL_STAMP(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < TAP_PROF_WGS) \
        tap_prof_tl[(blockIdx.x * TAP_PROF_WAVES + (threadIdx.x >> 6)) * 4 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define TL_WAIT_VM() __builtin_amdgcn_s_waitcnt(0)     /* every outstanding load / store of the wave has completed */
#else
#define TL_STAMP(i) do { } while (0)
#define TL_WAIT_VM() do { } while (0)
#endif

// LDS hand-off between lanes of ONE wavefront (same as tap_wave_lds_sync in tap_place.h)
__device__ __forceinline__ void tap_wave_lds_sync_m()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 16-byte store of a tensor this kernel will not touch again, in one of two flavours chosen per launch (`wt`,
// wave-uniform, tap_write_through() below):
//   write-through (`sc0 sc1`)  A plain or nontemporal store leaves the line dirty in the XCD's L2, and what is still
//       dirty when the kernel ends is written back at the kernel boundary, after the last wave -- at c2 most of the
//       19.7 MB a step writes (the eight L2s hold 32 MB).  A write-through store sends the bytes on while the kernel
//       is still running and drops the line.  Round 4, B = 8192 / 4096, per launch inside the replayed graph
//       (scripts/ab_transition.sh): c2 7.10-7.31 us nontemporal -> 6.70 (6.87 `sc1`, 8.03-8.12 plain, 8.69-8.86
//       `sc1 nt`), c3 8.80-8.84 -> 8.33-8.36; whole passes c2 1 139 -> 1 221 M env-steps/s, c3 458 -> 485 M, c4 441 ->
//       480 M, c6 125.5 -> 130.7 M.  Writing the step's SMALL outputs (masks, shadow words) through as well lost 4-8 %
//       at c2 (4-byte `sc1` stores are one fabric write each).
//   nontemporal  what round 1 chose against plain stores (+9 %); still the better one once a launch writes more than
//       the L2s hold, where the write-back overlaps the kernel anyway and write-through only adds fabric transactions.
//       Crossover (bench.py --sweep with TAP_WRITE_THROUGH=1 / 0, us per step): c2's shape 6.54 / 7.26 at B = 8 192
//       (19.7 MB), 12.5 / 13.0 at 16 384, 24.4 / 21.6 at 32 768 (78.6 MB), 65.5 / 38.3 at 65 536, 0.90 / 1.15 G
//       env-steps/s beyond; c3's shape 15.5 / 17.5 at 8 192 (59 MB), 37.0 / 30.8 at 16 384 -> write-through up to
//       TAP_WT_MAX_BYTES = 64 MB per launch.  (The MACS step, whose launches last 17 us and more, still gains at c4's
//       78.6 MB -- 480 against 441 M env-steps/s -- and takes a higher limit, transition_macs.hip.)
__device__ __forceinline__ void store_stream(float4 *dst, const float4 &v, int wt)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f nv = {v.x, v.y, v.z, v.w};
    // The two wait states behind the store belong to it: a VMEM store of more than 8 bytes reads its data registers AFTER it
    // issues, and a vector instruction that overwrites one of them right behind it changes what is stored (2 wait states on
    // gfx940+).  The compiler's hazard recogniser inserts them for the stores it emits itself -- it does not look inside
    // inline assembly.  Found in round 6 when a register allocation put the run's row counter (rr += RP) in the first data
    // register: element 0 of a row's float4 held the integer r + RP (test_fused_transition_equals_two_launch_path[64x64]).
    if (wt) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(dst), "v"(nv) : "memory");
    else __builtin_nontemporal_store(nv, reinterpret_cast<v4f *>(dst));
}

// Which flavour a launch that writes `bytes` of expanded fp32 tensor takes.  TAP_WRITE_THROUGH=0|1 in the
// environment forces one (A/B runs; read once).
inline int tap_write_through(size_t bytes, size_t limit)
{
    static const int forced = [] { const char *e = getenv("TAP_WRITE_THROUGH"); return e ? (e[0] != '0') : -1; }();
    if (forced >= 0) return forced;
    return bytes <= limit;
}
inline int tap_write_through(size_t bytes) { return tap_write_through(bytes, TAP_WT_MAX_BYTES); }

// bit r of a 64-bit column word as 0.f / 1.f without a 64-bit variable shift (quarter rate on CDNA)
__device__ __forceinline__ float bit_as_float(unsigned long long w, int r)
{
    const unsigned half = r < 32 ? (unsigned)w : (unsigned)(w >> 32);
    return (float)((half >> (r & 31)) & 1u);
}

// `while ptr >= n: ptr -= n` (pack.py:314-316) for a caller-provided column index: a valid index
// (< nR = n*R) needs at most R - 1 subtractions; an invalid one (the reference raises) is mapped to
// column 0 first instead of looping 2^60 times
__device__ __forceinline__ long tap_mod_col(long p, int n, int nR)
{
    if ((unsigned long)p >= (unsigned long)nR) p = 0;
    while (p >= n) p -= n;
    return p;
}

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
    // an index outside [0, nR) (the reference's gather raises) selects nothing: no row is cleared
    // (real = -1 from the caller) and no column leaves the mask
    const long real_m = (p >= 0 && p < nR) ? tap_mod_col(p, a.n, a.nR) : -1 - (long)a.n * a.R; // pack.py:314-316
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
                float r0 = -1.f;                                              // pack.py:339 via shuffle;
#pragma unroll                                                                // stays -1 for an index outside [0, nR)
                for (int c = 0; c < NC; ++c) {
                    const float t = __shfl(row0[k][c], (int)(p[k] & 63));
                    if ((p[k] >> 6) == c && p[k] >= 0 && p[k] < nR) r0 = t;
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
                d4[q] = v[u]; // regular store: the next step reads this tensor back (nontemporal: -16 %)
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
        const long real_m = (p[k] >= 0 && p[k] < nR) ? tap_mod_col(p[k], a.n, a.nR) : -1 - (long)a.n * a.R; // pack.py:314-316
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

// ---- the same step on the bit shadow ---------------------------------------------------------------
// `dynamic` only ever holds 0 and 1 (precedence matrices, pack.py:101-195), so the step can carry it
// as rows <= 64 bits per column: clearing the chosen rows is one AND per column, the three column
// sums of pack.py:323-326 are popcounts, and the fp32 tensor the network consumes (model.py:378) is
// EXPANDED from the bits instead of copied -- the step writes `rows*nR*4` bytes per env and reads
// `nR*8`, half the traffic of the copy.  One round trip: every input is loaded up front and `real`
// comes out of the row-0 registers by shuffle, as above.  (The expansion as a nibble -> float4 table look-up,
// which took a third of the vector instructions out of the rolling step (rolling.hip), was measured here too:
// 7 % SLOWER at the BASELINE batches -- the LDS read sits in every store's dependency chain of a latency-bound
// launch -- and +8 % only from B = 524 288; not kept.)  Lane (rsub, c4) = lane / (nR/4), lane %
// (nR/4) keeps four column words and writes the float4 of rows rsub, rsub + 64/(nR/4), ...: a wave
// store instruction covers whole consecutive rows.  Requirements: nR % 4 == 0, nR <= 64*NC, rows <= 64.
// First step of an episode (mask_builds_bits): the column words come from the fp32 slab instead of a stored
// shadow.  Lane (rsub, c4) reads the float4 of rows rsub, rsub + RP, ... (the mapping it writes with, all
// loads in flight together), packs "element != 0" into four partial words and ORs them into a wave-private
// LDS tile of nR words (ds_or_b64); after the wave-level hand-off every lane picks up the words it needs.
template <int NS>
__device__ __forceinline__ void stream_build_bits(const MaskArgs &a, int senv0, int lane, const bool (&on)[NS],
                                                  unsigned long long *tile /* NS * nR words */)
{
    typedef unsigned long long u64;
    const int nR = a.nR, C4 = nR >> 2, rows = a.rows;
    const int rsub = (int)(((unsigned)lane * (unsigned)a.c4_magic) >> 16), c4 = lane - rsub * C4, RP = a.rp;
    for (int i = lane; i < NS * nR; i += 64) tile[i] = 0ull;
    tap_wave_lds_sync_m();
    int bad = 0;
    // rows <= 64 and RP >= 1: at most 64 / RP row groups; the common shapes (30 rows: RP = 12 or 4) need 3 or 8
    // loads per lane and slab -- all of a wave's loads (both slabs) are issued before the first is used
    constexpr int U = 4;
    for (int r0 = rsub; r0 < rows; r0 += U * RP) {
        float4 v[NS][U];
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const float4 *src = reinterpret_cast<const float4 *>(a.dyn_in + (size_t)(senv0 + (on[k] ? k : 0)) * rows * nR) + c4;
#pragma unroll
            for (int u = 0; u < U; ++u) v[k][u] = src[(size_t)min(r0 + u * RP, rows - 1) * C4];
        }
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            unsigned lo[4] = {0u, 0u, 0u, 0u}, hi[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * RP;
                if (r >= rows || rsub >= RP || !on[k]) continue;
                const float e[4] = {v[k][u].x, v[k][u].y, v[k][u].z, v[k][u].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned nz = e[q] != 0.f;
                    bad += (nz && e[q] != 1.f);
                    if (r < 32) lo[q] |= nz << r; else hi[q] |= nz << (r - 32);
                }
            }
            if (on[k] && rsub < RP) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u64 word = ((u64)hi[q] << 32) | lo[q];
                    if (word) atomicOr(&tile[k * nR + c4 * 4 + q], word);
                }
            }
        }
    }
    if (a.nonbinary) {
        const unsigned long long any = __ballot(bad != 0);
        if (any) {                                               // rare: not a 0/1 tensor
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
            if (lane == 0) atomicAdd(a.nonbinary, bad);
        }
    }
    tap_wave_lds_sync_m();
}

// BUILD: the words come from the wave's LDS tile (first step, mask_builds_bits) instead of a.bits_in; a
// compile-time switch so that neither form reads through a generic pointer.
//
// Round 4, from the ISA of the round-3 form: (1) its input loads sat behind `cond ? ptr[i] : 0` selects, which the
// compiler turns into exec-masked regions it will not move loads across -- the shadow words were only requested after
// `ptr` / row 0 / the old mask had ARRIVED (an s_waitcnt vmcnt(0) between the two groups: two round trips, not one);
// (2) the expansion loops have run-time trip counts, so at the next use of a loaded register the wait-count pass no
// longer knows how many stores are outstanding and waits for vmcnt(0) -- the second slab's expansion started only after
// every store of the first had been ACKNOWLEDGED.  Now every input of both slabs is loaded unconditionally (clamped
// index, absent inputs read a valid dummy address, values selected afterwards), the wave waits ONCE for all of them,
// and nothing after that wait reads a register a load is still writing: the stores of both slabs go out back to back.
__device__ __forceinline__ int tap_mod_small(int v, int n)     // v mod n for v < 6 n (R <= 6 rotations), any v >= 0 otherwise
{
#pragma unroll
    for (int r = 0; r < 5; ++r) v = v >= n ? v - n : v;
    while (v >= n) v -= n;
    return v;
}

// (Round 4, measured and dropped: the fp32 tensor expanded LINEARLY over a wave's slabs -- float4 number i * 64 + lane of
//  the wave's contiguous region, the cleared column words fetched from a wave-private LDS tile -- so that every store
//  instruction covers 1 KiB of whole 128-byte lines instead of the 960 / 960 / 480-byte pieces of the (rsub, c4) mapping.
//  The bare store shapes say it should pay (tap_bw_probe kinds 6 / 7 at 19.66 MB: 4.51 against 4.35 us on a
//  cache-resident buffer, 6.14 against 4.02 us on fresh ones, profiles/r04_bw_store_shapes.json); the kernel did not
//  agree: c2 1 295 -> 1 268 M env-steps/s, c3 520 -> 478 M, cold passes 1 009 -> 992 M, and 10-15 % slower from
//  B = 128 k up -- the LDS read and the index arithmetic in front of every store cost more than the half lines.)
// C4S != 0: a BASELINE window with its shape known at compile time -- C4S = 5 / 15: n = 10 nodes, rows = 30, nR = 20 (2D) /
// 60 (3D) columns; C4S = 10: n = 20, rows = 60, nR = 40 (c4's 2D window) --: the expansion's loop unrolls, the row / column
// arithmetic folds.  Same session, same tree: c2 1 280 -> 1 337-1 346 M env-steps/s, c3 500 -> 506 M, c4 496 -> 504 M.
// INPLACE (never with BUILD): a.dyn_out holds the previous step's tensor -- only the rows this step clears are written
// FULL: every input is there (ptr, static, mask_in) and every slab of every wave is an env (B is a multiple of the envs per
// workgroup) -- the launcher checks; the wave then carries no code for the absent cases (the initial mask, a stepper's
// first step, the ragged last workgroup): ~40 of its ~550 instructions
template <int NS, int NC, bool BUILD = false, bool MERGED = true, int C4S = 0, bool INPLACE = false, bool FULL = false>
__device__ __forceinline__ void stream_wave_bits_r4(const MaskArgs &a, int senv0, int lane, const bool (&on_)[NS],
                                                 float *lds = nullptr)
{
    static_assert(!(BUILD && INPLACE), "the first step on a fresh tensor writes all of it");
    bool on[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) on[k] = FULL ? true : on_[k];
    // One load instruction fewer in the wave's input burst (round 6): the per-column copy of the shadow words (lane j = column
    // j, for the masks at the end of the wave) is the same data as the quads the expansion reads (lane (row group, quad)).
    // Column j's NEW word sits in element j & 3 of the lane that plays role (0, j >> 2) = lane (j >> 2) + sb, so the tail
    // fetches it with four ds_bpermute per slab, behind the stores.  30-row windows only (the low word carries every row),
    // and one-slab waves only: same session, c3 549.6 -> 555.5 M env-steps/s, but c2 (two slabs: eight shuffles and their
    // selects in the tail) 1 499 -> 1 483 M although a timing build WITHOUT the second load and without a replacement read
    // + 2.2 % there.  -DTAP_STREAM_LOAD_BJ: round 5's second load everywhere; -DTAP_STREAM_BJX_ALL: the shuffles for two-slab waves too.
#if defined(TAP_STREAM_LOAD_BJ)
    constexpr bool BJ_X = false;
#elif defined(TAP_STREAM_BJX_ALL)
    constexpr bool BJ_X = !BUILD && NC == 1 && (C4S == 5 || C4S == 15);
#else
    constexpr bool BJ_X = !BUILD && NC == 1 && NS == 1 && (C4S == 5 || C4S == 15);
#endif
    typedef unsigned long long u64;
    static_assert(NS <= 2, "a stream wave expands one or two slabs");
    if (!on[0]) return;                                  // on[] is a prefix and wave-uniform: no env, nothing to do
    const int nR = C4S ? 4 * C4S : a.nR, C4 = C4S ? C4S : nR >> 2, rows = C4S ? (C4S == 10 ? 60 : 30) : a.rows, n = C4S ? (C4S == 10 ? 20 : 10) : a.n;
    const int RP = C4S ? 64 / C4S : a.rp;
    const int c4_magic = C4S ? 65536 / C4S + 1 : a.c4_magic;
    const bool lane_on = lane < RP * C4;
    // ONE role per wave: the wave's slabs are contiguous, so their rows form one run of NS * rows rows of nR floats; the
    // lane keeps column quad c4 of EVERY slab and walks the rows r0, r0 + RP, ... of the whole run (stream_lane_role:
    // rotated so that each store instruction starts on a 64-byte granule).  Same registers as the fixed (rsub, c4)
    // mapping of round 4 -- per-slab roles cost six more and a wave per SIMD (68 VGPRs, measured: 1 264 against
    // 1 450 M env-steps/s at B = 1 M).
#ifdef TAP_STREAM_UNALIGNED   // A/B builds: round 4's fixed roles (store instructions start wherever the slab does)
    const LaneRole role = stream_lane_role(lane, 0, C4, RP, c4_magic, 0, 0, 0);
#else
    const LaneRole role = stream_lane_role(lane, senv0, C4, RP, c4_magic, a.sb_mul, a.sb_add, 0);
#endif
    const bool first_row = role.rsub == 0;               // these lanes also write the slab's new shadow words
    u64 *tile = reinterpret_cast<u64 *>(lds);
    // a readable, 16-byte aligned address for the inputs a caller may leave out (ptr / static: the initial mask,
    // model.py:297-307; mask_in: a stepper's first step starts from ones)
    const void *any = BUILD ? static_cast<const void *>(a.dyn_in) : static_cast<const void *>(a.bits_in);
    const bool has_ptr = FULL || a.ptr != nullptr, has_static = FULL || a.static_ != nullptr, has_mask = FULL || a.mask_in != nullptr;
    const int64_t *ptrp = has_ptr ? a.ptr : static_cast<const int64_t *>(any);
    const float *stp = has_static ? a.static_ : static_cast<const float *>(any);
    const float *mip = has_mask ? a.mask_in : static_cast<const float *>(any);
    long praw[NS];
    float row0[NS][NC], keep[NS][NC];
    u64 bj[NS][NC];
    ulonglong2 w[NS][2];
    // Addresses as (per-env base, small lane offset): with the env in a scalar register (TAP_WAVE_INDEX, the fused 2D step) the
    // bases are scalar arithmetic and a load costs the vector ALU one instruction instead of a 64-bit multiply-add, two
    // selects and a shift-add
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int env = on[k] ? senv0 + k : senv0;       // an idle slab re-reads the first one's inputs
        praw[k] = ptrp[has_ptr ? env : 0];
        const float *strow = stp + (has_static ? (size_t)env * a.static_rows * nR : (size_t)0);
        const float *mirow = mip + (has_mask ? (size_t)env * nR : (size_t)0);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const unsigned j = (unsigned)min(lane + 64 * c, nR - 1);
            row0[k][c] = strow[has_static ? j : 0u];
            keep[k][c] = mirow[has_mask ? j : 0u];
        }
    }
#ifdef TAP_STREAM_G2          // A/B builds: the shadow words requested only after the small inputs have arrived (round 3's order)
    __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
    if (!BUILD) {
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int env = on[k] ? senv0 + k : senv0;
            const unsigned long long *brow = a.bits_in + (size_t)env * nR;
            // (BJ_X: the compiled-in 10-node windows take a column's word from the lane that holds its quad, after the
            //  clear -- below -- instead of loading the shadow a second time in the per-column layout)
            if constexpr (!BJ_X) {
#pragma unroll
                for (int c = 0; c < NC; ++c) bj[k][c] = brow[(unsigned)min(lane + 64 * c, nR - 1)];
            }
            const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(brow) + (unsigned)(role.c4 * 2);
            w[k][0] = src[0];
            w[k][1] = src[1];
        }
    }
    if (BUILD) {
        stream_build_bits<NS>(a, senv0, lane, on, tile);
#pragma unroll
        for (int k = 0; k < NS; ++k) {
#pragma unroll
            for (int c = 0; c < NC; ++c) bj[k][c] = tile[(size_t)k * nR + min(lane + 64 * c, nR - 1)];
            const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(tile + (size_t)k * nR + role.c4 * 4);
            w[k][0] = src[0];
            w[k][1] = src[1];
        }
    }
    // The OUTPUT addresses and the store flavour live in the argument block and the compiler fetches each at its first use,
    // AFTER the wait below (three scalar-cache reads inside the wave's critical path).  Forcing them into scalar registers
    // here, while the vector loads are in flight, measured SLOWER (round 6, -DTAP_STREAM_EARLY_ARGS: c2 1 488 -> 1 450 M
    // env-steps/s with the rolled expansion, 1 502 -> 1 464 M with the unrolled one): the wait it puts in front of the
    // vector wait costs more than the late reads, which hit the scalar cache.
    float *o_dyn = a.dyn_out, *o_cur = a.cur_out, *o_mask = a.mask_out;
    unsigned long long *o_bits = a.bits_out;
    int o_wt = a.wt;
#ifdef TAP_STREAM_EARLY_ARGS                              // A/B builds
    asm volatile("" : "+s"(o_dyn), "+s"(o_cur), "+s"(o_mask), "+s"(o_bits), "+s"(o_wt));
#endif
    __builtin_amdgcn_s_waitcnt(0x0F70);                  // vmcnt(0): every input of both slabs is here
#ifdef TAP_PROF
    TL_STAMP(1);
#endif
    const u64 nmask = (n >= 64) ? ~0ull : ((1ull << n) - 1ull);
    // v mod n for a column index (pack.py:314-316): a division by a constant for the compiled-in windows
    auto mod_n = [&](int v) -> int {
        if constexpr (C4S != 0) return (int)((unsigned)v % (unsigned)(C4S == 10 ? 20 : 10));
        else return tap_mod_small(v, n);
    };
    int jm[NC];                                                               // this lane's columns mod n (pack.py:314-316 for a column)
#pragma unroll
    for (int c = 0; c < NC; ++c) jm[c] = mod_n(min(lane + 64 * c, nR - 1));
    u64 clr[NS];
    int pm[NS], realk[NS];
    // A slab's pick, its block id and the rows it clears are the same on every lane of the wave: kept in SCALAR registers
    // (v_readlane on the row-0 values the lanes hold instead of a shuffle through the LDS crossbar; the clear mask is
    // scalar shifts), they cost the stream wave's critical path -- this launch's, at the BASELINE batches -- no vector
    // issue slots and no LDS round trip (round 6: c2 1 440 -> 1 470 M env-steps/s, scripts/ab_transition.sh)
    constexpr int UR_C = C4S != 0 ? 3 : 0;                 // the compiled-in windows are 'bot' windows: rows = 3 n, update_rows = 3 (checked by the launchers)
    const int ur = UR_C ? UR_C : a.update_rows;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const bool valid_v = on[k] && has_ptr && praw[k] >= 0 && praw[k] < nR;  // no ptr: the initial mask (model.py:297-307)
        // (two-slab waves only: with ONE slab per wave -- 3D windows, c3 -- the vector form below measured 1 % faster,
        //  538 against 543 M env-steps/s: there the chain is half as long and the scalar unit is the busier one)
#ifdef TAP_STREAM_VPICK                                     // A/B builds: round 5's vector form everywhere
        constexpr bool SPICK = false;
#else
        constexpr bool SPICK = NS > 1;
#endif
        bool valid;
        int p, real;
        if constexpr (!SPICK) {
            valid = valid_v;
            p = valid ? (int)praw[k] : 0;
            float r0 = -1.f;                                                  // pack.py:339 via shuffle; stays -1 for an index outside [0, nR)
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float t = __shfl(row0[k][c], p & 63);
                if (valid && has_static && (p >> 6) == c) r0 = t;
            }
            real = (r0 > -1.f && r0 < (float)rows) ? (int)r0 : -1;            // .long() truncates; a row beyond the tensor clears nothing
        } else {
            const int ps = __builtin_amdgcn_readfirstlane(valid_v ? (int)praw[k] : -1);
            valid = ps >= 0;
            p = valid ? ps : 0;
            float r0 = -1.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(row0[k][c]), p & 63));
                if (valid && has_static && (p >> 6) == c) r0 = t;
            }
            real = __builtin_amdgcn_readfirstlane((r0 > -1.f && r0 < (float)rows) ? (int)r0 : -1);
        }
        u64 m = 0;                                                            // pack.py:370-374
#pragma unroll
        for (int i = 0; i < 3; ++i) {                                         // update_rows <= 3 (validated on the host)
            const int r = real + n * i;
            if (i < ur && real >= 0 && r < rows) m |= 1ull << r;
        }
        clr[k] = m;
        realk[k] = real;
        pm[k] = valid ? mod_n(p) : -1;                                        // pack.py:314-316; -1 matches no column
    }
    // the new words of this lane's column quad, per slab
    u64 nw[NS][4];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        nw[k][0] = w[k][0].x & ~clr[k]; nw[k][1] = w[k][0].y & ~clr[k];
        nw[k][2] = w[k][1].x & ~clr[k]; nw[k][3] = w[k][1].y & ~clr[k];
    }
    if constexpr (INPLACE) {
        // pack.py:370-374 on a tensor that is already there: rows real + n * i of the slab become zeros.  Lane = (i, column
        // quad): ONE store instruction per slab for windows of up to 21 columns per rotation (3 * nR / 4 <= 64 lanes); the
        // stream wave is this step's critical path and every instruction in front of its last store is paid in full
        // (a row-by-row loop, three instructions per slab behind two nested loops: 5.60 against 5.20 us per launch at c2
        // with the tensor switched off)
        if (o_dyn) {
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const int total = ur * C4;
            for (int idx = lane; idx < total; idx += 64) {
                const int i = (idx * c4_magic) >> 16, c = idx - i * C4;          // idx / C4 (idx < 3 * 64)
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const int r = realk[k] + n * i;
                    if (!on[k] || realk[k] < 0 || r >= rows) continue;
                    float4 *dst = reinterpret_cast<float4 *>(o_dyn) + ((size_t)(senv0 + k) * rows + r) * C4 + c;
#if defined(TAP_INPLACE_STORE) && TAP_INPLACE_STORE == 1      // A/B builds: always write-through / nontemporal like the expansion
                    store_stream(dst, z, o_wt);
#elif defined(TAP_INPLACE_STORE) && TAP_INPLACE_STORE == 2     // A/B builds: plain
                    *dst = z;
#else
                    if (o_wt) store_stream(dst, z, 1); else *dst = z;            // a row is a fraction of a line: never nontemporal
#endif
                }
            }
        }
    }
    if (lane_on) {
        if (!INPLACE && o_dyn) {
            // the fp32 tensor: instruction i = the lane's row r0 + RP * i of the wave's run when that is a row of it (late
            // lanes skip i = 0); rows >= `rows` belong to the second slab
            const int non = (NS > 1 && on[NS - 1]) ? 2 : 1;
            const int total = non * rows;
            const int nq = (C4S ? (non * rows + RP - 1) / RP : (non > 1 ? a.nq2 : a.nq)) + role.nq;   // role.nq: one more instruction when rotated
            float4 *dst = reinterpret_cast<float4 *>(o_dyn + (size_t)senv0 * rows * nR) + role.c4;
            int rr = role.r0;
            // MERGED = false (two-slab waves only): round 4's loops, one per slab on the lane's own rows.  Which form a
            // kernel takes is decided where it is launched (transition.hip), from these A/B runs (env-steps/s, same
            // session, profiles/r05_store_shape_ab.txt): the run-of-rows loop c2 1 279 against 1 258 M and c4 504 against
            // 490 M, but c3 495 against 509 M (15 store instructions per run, the stream waves are that step's critical
            // path) and, wherever the stores are nontemporal (the output no longer fits the caches), 1 549 against
            // 1 872 M at B = 128 k and 1 333-1 368 against 1 470-1 541 M at B = 1 M in c2's shape.  Compiling both forms
            // into one kernel behind a run-time switch cost the small-batch gain (1 256 M), hence the template.
            bool done = false;
            if constexpr (NS > 1 && !MERGED) {
                if (rows <= 32) {
#pragma unroll
                    for (int k = 0; k < NS; ++k) {
                        if (!on[k]) continue;
                        float4 *dk = dst + (size_t)k * rows * C4;
                        const unsigned m0 = (unsigned)nw[k][0], m1 = (unsigned)nw[k][1], m2 = (unsigned)nw[k][2], m3 = (unsigned)nw[k][3];
                        for (int r = role.rsub; r < rows; r += RP) {
                            const float4 v = make_float4((float)((m0 >> r) & 1u), (float)((m1 >> r) & 1u),
                                                         (float)((m2 >> r) & 1u), (float)((m3 >> r) & 1u));
                            store_stream(&dk[(size_t)r * C4], v, o_wt);
                        }
                    }
                    done = true;
                }
            }
            if (done) {
            } else if (C4S != 0 && rows <= 32) {
                // the compiled-in 10-node windows: the run's instructions unrolled (at most NQC of them; the guard skips the
                // ones a lane does not have), the store flavour chosen once instead of inside the loop (round 6: c2 1 488 ->
                // 1 502 M env-steps/s, c3 unchanged)
                constexpr int ROWS_C = 30, RP_C = C4S ? 64 / C4S : 1, NQC = (NS * ROWS_C + RP_C - 1) / RP_C + 1;
                auto run = [&](auto wt_c) {
#pragma unroll
                    for (int i = 0; i < NQC; ++i) {
                        const int ri = rr + RP_C * i;
                        if ((unsigned)ri >= (unsigned)total) continue;
                        const bool up = NS > 1 && ri >= ROWS_C;
                        const int r = up ? ri - ROWS_C : ri;
                        const unsigned m0 = up ? (unsigned)nw[NS - 1][0] : (unsigned)nw[0][0], m1 = up ? (unsigned)nw[NS - 1][1] : (unsigned)nw[0][1],
                                       m2 = up ? (unsigned)nw[NS - 1][2] : (unsigned)nw[0][2], m3 = up ? (unsigned)nw[NS - 1][3] : (unsigned)nw[0][3];
                        const float4 v = make_float4((float)((m0 >> r) & 1u), (float)((m1 >> r) & 1u),
                                                     (float)((m2 >> r) & 1u), (float)((m3 >> r) & 1u));
                        store_stream(&dst[(size_t)ri * C4], v, decltype(wt_c)::value);
                    }
                };
#ifdef TAP_STREAM_ROLLED                                  // A/B builds: round 5's rolled loop
                for (int i = 0; i < nq; ++i, rr += RP) {
                    if ((unsigned)rr >= (unsigned)total) continue;
                    const bool up = NS > 1 && rr >= rows;
                    const int r = up ? rr - rows : rr;
                    const unsigned m0 = up ? (unsigned)nw[NS - 1][0] : (unsigned)nw[0][0], m1 = up ? (unsigned)nw[NS - 1][1] : (unsigned)nw[0][1],
                                   m2 = up ? (unsigned)nw[NS - 1][2] : (unsigned)nw[0][2], m3 = up ? (unsigned)nw[NS - 1][3] : (unsigned)nw[0][3];
                    const float4 v = make_float4((float)((m0 >> r) & 1u), (float)((m1 >> r) & 1u),
                                                 (float)((m2 >> r) & 1u), (float)((m3 >> r) & 1u));
                    store_stream(&dst[(size_t)rr * C4], v, o_wt);
                }
#else
                if (o_wt) run(std::integral_constant<int, 1>{}); else run(std::integral_constant<int, 0>{});
#endif
            } else if (rows <= 32) {
                // n <= 10 (every BASELINE window): the column words fit 32 bits -- a bit-field extract and a
                // convert per element, no half selection
                for (int i = 0; i < nq; ++i, rr += RP) {
                    if ((unsigned)rr >= (unsigned)total) continue;
                    const bool up = NS > 1 && rr >= rows;
                    const int r = up ? rr - rows : rr;
                    const unsigned m0 = up ? (unsigned)nw[NS - 1][0] : (unsigned)nw[0][0], m1 = up ? (unsigned)nw[NS - 1][1] : (unsigned)nw[0][1],
                                   m2 = up ? (unsigned)nw[NS - 1][2] : (unsigned)nw[0][2], m3 = up ? (unsigned)nw[NS - 1][3] : (unsigned)nw[0][3];
                    const float4 v = make_float4((float)((m0 >> r) & 1u), (float)((m1 >> r) & 1u),
                                                 (float)((m2 >> r) & 1u), (float)((m3 >> r) & 1u));
                    store_stream(&dst[(size_t)rr * C4], v, o_wt);
                }
            } else {
                for (int i = 0; i < nq; ++i, rr += RP) {
                    if ((unsigned)rr >= (unsigned)total) continue;
                    const bool up = NS > 1 && rr >= rows;
                    const int r = up ? rr - rows : rr;
                    const float4 v = make_float4(bit_as_float(up ? nw[NS - 1][0] : nw[0][0], r), bit_as_float(up ? nw[NS - 1][1] : nw[0][1], r),
                                                 bit_as_float(up ? nw[NS - 1][2] : nw[0][2], r), bit_as_float(up ? nw[NS - 1][3] : nw[0][3], r));
                    store_stream(&dst[(size_t)rr * C4], v, o_wt);
                }
            }
        }
        if (first_row && o_bits) {
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                if (!on[k]) continue;
                ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(o_bits + (size_t)(senv0 + k) * nR + role.c4 * 4);
                dst[0] = make_ulonglong2(nw[k][0], nw[k][1]);
                dst[1] = make_ulonglong2(nw[k][2], nw[k][3]);
            }
        }
    }
    unsigned nbx[NS];                                     // BJ_X: column `lane`'s new word, from the lane that holds its quad
    if constexpr (BJ_X) {
        const int jj = min(lane, nR - 1);
        const int src = ((jj >> 2) + role.sb) << 2;       // ds_bpermute addresses lanes in bytes; every lane of the wave takes part
        const int e = jj & 3;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const int e0 = __builtin_amdgcn_ds_bpermute(src, (int)(unsigned)nw[k][0]), e1 = __builtin_amdgcn_ds_bpermute(src, (int)(unsigned)nw[k][1]);
            const int e2 = __builtin_amdgcn_ds_bpermute(src, (int)(unsigned)nw[k][2]), e3 = __builtin_amdgcn_ds_bpermute(src, (int)(unsigned)nw[k][3]);
            nbx[k] = (unsigned)(e == 0 ? e0 : e == 1 ? e1 : e == 2 ? e2 : e3);
        }
    }
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        if (!on[k]) continue;
        const int env = senv0 + k;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = lane + 64 * c;
            if (j >= nR) continue;
            const u64 nb = BJ_X ? (u64)nbx[k] : (bj[k][c] & ~clr[k]);
            const int move = __popcll(nb & nmask), small = n >= 64 ? 0 : __popcll((nb >> n) & nmask);
            const int large = n >= 32 ? 0 : __popcll((nb >> (2 * n)) & nmask); // rows <= 64
            const float kp = (jm[c] == pm[k]) ? 0.f : (has_mask ? keep[k][c] : 1.f);   // pack.py:320-321
            if (o_mask) o_mask[(size_t)env * nR + j] = kp;
            if (o_cur) o_cur[(size_t)env * nR + j] = (small * large + move) != 0 ? 0.f : kp; // :327-329
        }
    }
}

// (Round 5, measured and removed: a stream wave walking 2 or 4 consecutive slab pairs with the NEXT pair's inputs put in
//  flight right behind the current pair's stores, for launches beyond the write-through limit -- the load round trip of
//  pair i + 1 overlapping the drain of pair i inside the wave, one vmcnt(0) per pair.  c2's shape: 1 628 / 1 612 against
//  1 653 M env-steps/s at B = 128 k, 1 353 against 1 471 M at 512 k, 1 357 / 1 361 against 1 478 M at 1 M; c3's at 256 k
//  514 / 500 against 558 M.  With every wave slot of the CU taken, the hardware's interleaving of many short-lived waves
//  already overlaps loads and drains; fewer, longer-lived waves only lose.  git: "pipelined stream wave ... TAP_PIPE_ITERS".)
// -DTAP_STREAM_R3: the round-3 form of the step (scripts/ab_transition.sh A/B builds only)
#ifdef TAP_STREAM_R3
template <int NS, int NC, bool BUILD = false>
__device__ __forceinline__ void stream_wave_bits_r3(const MaskArgs &a, int senv0, int lane, const bool (&on)[NS],
                                                 float *lds = nullptr)
{
    typedef unsigned long long u64;
    const int nR = a.nR, C4 = nR >> 2, rows = a.rows;
    const int rsub = (int)(((unsigned)lane * (unsigned)a.c4_magic) >> 16), c4 = lane - rsub * C4, RP = a.rp;
    const bool lane_on = rsub < RP;
    constexpr bool build = BUILD;
    u64 *tile = reinterpret_cast<u64 *>(lds);
    long p[NS];
    float row0[NS][NC], keep[NS][NC];
    u64 bj[NS][NC];
    ulonglong2 w[NS][2];
    // every small input is requested up front (in the first-step form: before the slab is read, so the
    // step still has one memory round trip)
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int env = senv0 + k;
        p[k] = (on[k] && a.ptr) ? (long)a.ptr[env] : -1;   // no ptr: the initial mask (model.py:297-307)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = lane + 64 * c;
            const bool ok = on[k] && j < nR;
            row0[k][c] = (ok && a.static_) ? a.static_[(size_t)env * a.static_rows * nR + j] : 0.f;
            keep[k][c] = ok ? (a.mask_in ? a.mask_in[(size_t)env * nR + j] : 1.f) : 0.f;
        }
    }
    if (BUILD) stream_build_bits<NS>(a, senv0, lane, on, tile);
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        const int env = senv0 + k;
        const size_t wbase = build ? (size_t)k * nR : (size_t)env * nR;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = lane + 64 * c;
            const bool ok = on[k] && j < nR;
            if (BUILD) bj[k][c] = ok ? tile[wbase + j] : 0ull;
            else bj[k][c] = ok ? a.bits_in[wbase + j] : 0ull;
        }
        const bool ok = on[k] && lane_on;
        if (BUILD) {
            const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(tile + wbase + c4 * 4);
            w[k][0] = ok ? src[0] : make_ulonglong2(0, 0);
            w[k][1] = ok ? src[1] : make_ulonglong2(0, 0);
        } else {
            const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(a.bits_in + wbase + c4 * 4);
            w[k][0] = ok ? src[0] : make_ulonglong2(0, 0);
            w[k][1] = ok ? src[1] : make_ulonglong2(0, 0);
        }
    }
    const u64 nmask = (a.n >= 64) ? ~0ull : ((1ull << a.n) - 1ull);
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        if (!on[k]) continue;
        const int env = senv0 + k;
        float r0 = -1.f;                                                      // pack.py:339 via shuffle; stays -1
#pragma unroll                                                                        // for an index outside [0, nR)
        for (int c = 0; c < NC; ++c) {
            const float t = __shfl(row0[k][c], (int)(p[k] & 63));
            if ((p[k] >> 6) == c && p[k] >= 0 && p[k] < nR) r0 = t;
        }
        const long real = (long)r0;
        u64 clr = 0;                                                          // pack.py:370-374
        for (int i = 0; i < a.update_rows; ++i) {
            const long r = real + (long)a.n * i;
            if (real >= 0 && r < rows) clr |= 1ull << r;
        }
        if (lane_on) {
            const u64 n0 = w[k][0].x & ~clr, n1 = w[k][0].y & ~clr, n2 = w[k][1].x & ~clr, n3 = w[k][1].y & ~clr;
#ifdef TAP_PROF
            if (k == 0) { TL_WAIT_VM(); TL_STAMP(1); }
#endif
            if (a.dyn_out) {
                float4 *dst = reinterpret_cast<float4 *>(a.dyn_out + (size_t)env * rows * nR) + c4;
                if (rows <= 32) {
                    // n <= 10 (every BASELINE window): the column words fit 32 bits -- a bit-field extract and a
                    // convert per element, no half selection
                    const unsigned m0 = (unsigned)n0, m1 = (unsigned)n1, m2 = (unsigned)n2, m3 = (unsigned)n3;
                    for (int r = rsub; r < rows; r += RP) {
                        const float4 v = make_float4((float)((m0 >> r) & 1u), (float)((m1 >> r) & 1u),
                                                     (float)((m2 >> r) & 1u), (float)((m3 >> r) & 1u));
                        store_stream(&dst[(size_t)r * C4], v, a.wt);
                    }
                } else {
                    for (int r = rsub; r < rows; r += RP) {
                        const float4 v = make_float4(bit_as_float(n0, r), bit_as_float(n1, r), bit_as_float(n2, r),
                                                     bit_as_float(n3, r));
                        store_stream(&dst[(size_t)r * C4], v, a.wt);
                    }
                }
            }
            if (rsub == 0 && a.bits_out) {
                ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(a.bits_out + (size_t)env * nR + c4 * 4);
                dst[0] = make_ulonglong2(n0, n1);
                dst[1] = make_ulonglong2(n2, n3);
            }
        }
        const long real_m = (p[k] >= 0 && p[k] < nR) ? tap_mod_col(p[k], a.n, a.nR) : -1 - (long)a.n * a.R; // pack.py:314-316
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = lane + 64 * c;
            if (j >= nR) continue;
            const u64 nb = bj[k][c] & ~clr;
            const int move = __popcll(nb & nmask), small = a.n >= 64 ? 0 : __popcll((nb >> a.n) & nmask);
            const int large = a.n >= 32 ? 0 : __popcll((nb >> (2 * a.n)) & nmask); // rows <= 64
            float kp = keep[k][c];
            for (int r = 0; r < a.R; ++r)
                if (j == real_m + (long)a.n * r) kp = 0.f;                    // pack.py:320-321
            if (a.mask_out) a.mask_out[(size_t)env * nR + j] = kp;
            if (a.cur_out) a.cur_out[(size_t)env * nR + j] = (small * large + move) != 0 ? 0.f : kp; // :327-329
        }
    }
}

#endif
template <int NS, int NC, bool BUILD = false, bool MERGED = true, int C4S = 0, bool INPLACE = false, bool FULL = false>
__device__ __forceinline__ void stream_wave_bits(const MaskArgs &a, int senv0, int lane, const bool (&on)[NS], float *lds = nullptr)
{
#ifdef TAP_STREAM_R3
    stream_wave_bits_r3<NS, NC, BUILD>(a, senv0, lane, on, lds);
#else
    stream_wave_bits_r4<NS, NC, BUILD, MERGED, C4S, INPLACE, FULL>(a, senv0, lane, on, lds);
#endif
}

// ---- the bit shadow with TWO words per column: 65 <= rows <= 128 (windows of 22 .. 42 nodes) -------------
// Plane w of env b (rows 64w .. 64w + 63) lives at bits[(b*2 + w)*nR + j]; everything else is the step above
// (one round trip, `real` by shuffle, lane (rsub, c4) expands the float4 of rows rsub, rsub + RP, ...).  One
// slab per wave; lds: 2*nR words for the first-step form.
__host__ __device__ __forceinline__ int mask_bit_planes(int rows) { return rows > 64 ? 2 : 1; }

__device__ __forceinline__ unsigned long long bits_range64(int a, int b) // bits [a, b) clipped to [0, 64)
{
    a = a < 0 ? 0 : a;
    b = b > 64 ? 64 : b;
    if (a >= b) return 0ull;
    return ((b - a) >= 64 ? ~0ull : ((1ull << (b - a)) - 1ull)) << a;
}

template <int NC, bool BUILD>
__device__ __forceinline__ void stream_wave_bits2(const MaskArgs &a, int env, int lane, float *lds)
{
    typedef unsigned long long u64;
    const int nR = a.nR, C4 = nR >> 2, rows = a.rows;
    const int rsub = (int)(((unsigned)lane * (unsigned)a.c4_magic) >> 16), c4 = lane - rsub * C4, RP = a.rp;
    const bool lane_on = rsub < RP;
    u64 *tile = reinterpret_cast<u64 *>(lds);
    const long p = a.ptr ? (long)a.ptr[env] : -1;                            // no ptr: the initial mask
    float row0[NC], keep[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int j = lane + 64 * c;
        const bool ok = j < nR;
        row0[c] = (ok && a.static_) ? a.static_[(size_t)env * a.static_rows * nR + j] : 0.f;
        keep[c] = ok ? (a.mask_in ? a.mask_in[(size_t)env * nR + j] : 1.f) : 0.f;
    }
    if (BUILD) {
        for (int i = lane; i < 2 * nR; i += 64) tile[i] = 0ull;
        tap_wave_lds_sync_m();
        int bad = 0;
        constexpr int U = 4;
        const float4 *src = reinterpret_cast<const float4 *>(a.dyn_in + (size_t)env * rows * nR) + c4;
        u64 acc[2][4] = {{0ull, 0ull, 0ull, 0ull}, {0ull, 0ull, 0ull, 0ull}};
        for (int r0 = rsub; r0 < rows; r0 += U * RP) {
            float4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = src[(size_t)min(r0 + u * RP, rows - 1) * C4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * RP;
                if (r >= rows || !lane_on) continue;
                const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u64 nz = e[q] != 0.f;
                    bad += (nz && e[q] != 1.f);
                    if (r < 64) acc[0][q] |= nz << r; else acc[1][q] |= nz << (r - 64);
                }
            }
        }
        if (lane_on) {
#pragma unroll
            for (int w = 0; w < 2; ++w)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (acc[w][q]) atomicOr(&tile[w * nR + c4 * 4 + q], acc[w][q]);
        }
        if (a.nonbinary) {
            const unsigned long long any = __ballot(bad != 0);
            if (any) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o);
                if (lane == 0) atomicAdd(a.nonbinary, bad);
            }
        }
        tap_wave_lds_sync_m();
    }
    const u64 *win = BUILD ? tile : a.bits_in + (size_t)env * 2 * nR;
    const LaneRole role = stream_lane_role(lane, env, C4, RP, a.c4_magic, a.sb_mul, a.sb_add, a.nq);   // expansion: rotated roles
    u64 bj[NC][2], w4[2][4];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int j = lane + 64 * c;
#pragma unroll
        for (int w = 0; w < 2; ++w) bj[c][w] = j < nR ? win[w * nR + j] : 0ull;
    }
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        const ulonglong2 *src = reinterpret_cast<const ulonglong2 *>(win + w * nR + role.c4 * 4);
        const ulonglong2 s0 = lane_on ? src[0] : make_ulonglong2(0, 0), s1 = lane_on ? src[1] : make_ulonglong2(0, 0);
        w4[w][0] = s0.x; w4[w][1] = s0.y; w4[w][2] = s1.x; w4[w][3] = s1.y;
    }
    float r0 = -1.f;                                                          // pack.py:339 via shuffle
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float t = __shfl(row0[c], (int)(p & 63));
        if ((p >> 6) == c && p >= 0 && p < nR) r0 = t;
    }
    const long real = (long)r0;
    u64 clr[2] = {0ull, 0ull};                                                // pack.py:370-374
    for (int i = 0; i < a.update_rows; ++i) {
        const long r = real + (long)a.n * i;
        if (real >= 0 && r < rows) { if (r < 64) clr[0] |= 1ull << r; else clr[1] |= 1ull << (r - 64); }
    }
    if (lane_on) {
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int q = 0; q < 4; ++q) w4[w][q] &= ~clr[w];
        if (a.dyn_out) {
            float4 *dst = reinterpret_cast<float4 *>(a.dyn_out + (size_t)env * rows * nR) + role.c4;
            int r = role.r0;
            for (int i = 0; i < role.nq; ++i, r += RP) {
                if ((unsigned)r >= (unsigned)rows) continue;
                const bool up = r >= 64;
                const int rb = r & 63;
                const float4 v = make_float4(bit_as_float(up ? w4[1][0] : w4[0][0], rb), bit_as_float(up ? w4[1][1] : w4[0][1], rb),
                                             bit_as_float(up ? w4[1][2] : w4[0][2], rb), bit_as_float(up ? w4[1][3] : w4[0][3], rb));
                store_stream(&dst[(size_t)r * C4], v, a.wt);
            }
        }
        if (role.rsub == 0 && a.bits_out) {
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                ulonglong2 *dst = reinterpret_cast<ulonglong2 *>(a.bits_out + ((size_t)env * 2 + w) * nR + role.c4 * 4);
                dst[0] = make_ulonglong2(w4[w][0], w4[w][1]);
                dst[1] = make_ulonglong2(w4[w][2], w4[w][3]);
            }
        }
    }
    const long real_m = (p >= 0 && p < nR) ? tap_mod_col(p, a.n, a.nR) : -1 - (long)a.n * a.R;    // pack.py:314-316
    const int n = a.n;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int j = lane + 64 * c;
        if (j >= nR) continue;
        const u64 lo = bj[c][0] & ~clr[0], hi = bj[c][1] & ~clr[1];
        int sec[3];                                                           // pack.py:323-326 as popcounts
#pragma unroll
        for (int s = 0; s < 3; ++s)
            sec[s] = __popcll(lo & bits_range64(s * n, (s + 1) * n)) + __popcll(hi & bits_range64(s * n - 64, (s + 1) * n - 64));
        float kp = keep[c];
        for (int r = 0; r < a.R; ++r)
            if (j == real_m + (long)n * r) kp = 0.f;                          // pack.py:320-321
        if (a.mask_out) a.mask_out[(size_t)env * nR + j] = kp;
        if (a.cur_out) a.cur_out[(size_t)env * nR + j] = (sec[1] * sec[2] + sec[0]) != 0 ? 0.f : kp; // :327-329
    }
}

// the bit shadow needs 16-byte rows of words and float4 rows; rows <= 64: one word per column (every kernel that
// carries the step), 65 .. 128: two (masks.hip's stand-alone step; the fused entry points run it as their first launch)
inline bool mask_bits_ok(const MaskArgs &a)
{
    return (a.bits_in || mask_builds_bits(a)) && (a.ptr == nullptr || a.static_ != nullptr) && (a.nR % 4 == 0) &&
           a.nR <= 256 && a.rows >= 1 && a.rows <= 128 &&
           ((reinterpret_cast<uintptr_t>(a.dyn_out) | reinterpret_cast<uintptr_t>(a.dyn_in)) % 16 == 0) &&
           ((reinterpret_cast<uintptr_t>(a.bits_in) | reinterpret_cast<uintptr_t>(a.bits_out)) % 16 == 0);
}

// columns per lane the fast path needs: 1, 2 or 4; 0 = use the generic element-wise path
inline int mask_fast_path_cols(const MaskArgs &a)
{
    if (a.bits_in || mask_builds_bits(a)) return a.nR <= 64 ? 1 : a.nR <= 128 ? 2 : 4; // mask_bits_ok checked by the caller
    const bool ok = a.dyn_out && a.ptr && a.static_ && a.cs_in && (a.nR % 4 == 0) && a.nR <= 256 && a.rows >= 1 &&
                    ((reinterpret_cast<uintptr_t>(a.dyn_in) | reinterpret_cast<uintptr_t>(a.dyn_out)) % 16 == 0);
    return !ok ? 0 : a.nR <= 64 ? 1 : a.nR <= 128 ? 2 : 4;
}
