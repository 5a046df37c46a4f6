// ppsg.hip -- perfect-packing ("PPSG") instances on the device: generate.BPP_Generator_3D
// (generate.py:232-301), generate.BPP_Generator_2D_easy (generate.py:392-484) and the layout proposal /
// acceptance steps of generate.generate_blocks_with_GT (generate.py:57-161).  Instance generation is set-up work (not the per-step hot path): one thread per
// unit, scalar code, everything in registers / scratch.  gfx950 only.
//
// Randomness: the reference draws from numpy's RandomState; here every draw is made the way RandomState
// makes it (random_sample = 2 words -> 53 bits, randint = masked rejection, choice(p) = searchsorted on the
// normalised cumulative sum in fp64) but on a counter-based word stream: word i of stream `key` =
// hi32(splitmix64 finaliser(key + i * golden)).  oracle/tap_oracle.c implements the same streams, so the
// kernels are compared with the restatement that is itself pinned on the reference draw for draw.
#include "tap_common.h"

typedef unsigned long long u64;

__host__ __device__ static inline u64 ppsg_mix64(u64 z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__host__ __device__ static inline u64 ppsg_key(u64 seed, u64 a, u64 b, u64 c)
{
    u64 k = ppsg_mix64(seed + 0x9E3779B97F4A7C15ull);
    k = ppsg_mix64(k ^ (a + 0x9E3779B97F4A7C15ull));
    k = ppsg_mix64(k ^ (b + 0x9E3779B97F4A7C15ull));
    k = ppsg_mix64(k ^ (c + 0x9E3779B97F4A7C15ull));
    return k;
}

struct PpsgRng {
    u64 key, ctr;
    __device__ unsigned next32() { return (unsigned)(ppsg_mix64(key + (ctr++) * 0x9E3779B97F4A7C15ull) >> 32); }
    __device__ double sample()
    {
        const unsigned a = next32() >> 5, b = next32() >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
    __device__ long randint(long low, long high)
    {
        const u64 rng = (u64)(high - low - 1);
        if (rng == 0) return low;
        u64 mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        for (;;) {
            const u64 v = (u64)next32() & mask;
            if (v <= rng) return low + (long)v;
        }
    }
    // np.random.choice(range(k), p = w / sum(w)), integer weights (generate.py:252-253, 264-267)
    __device__ int choice_weighted(const long *w, int k)
    {
        long sum = 0;
        for (int i = 0; i < k; ++i) sum += w[i];
        double tot = 0.0;
        for (int i = 0; i < k; ++i) tot += (double)w[i] / (double)sum;   // cdf[-1]
        const double u = sample();
        double acc = 0.0;
        for (int i = 0; i < k; ++i) {                                    // searchsorted(cdf / cdf[-1], u, 'right')
            acc += (double)w[i] / (double)sum;
            if (acc / tot > u) return i;
        }
        return k - 1;
    }
};

constexpr int PPSG_MAX_SLAB = 16; // blocks per slab held in a thread's scratch

struct PpsgGtArgs {
    int B, S, ns, W, min_size, max_size, gen;
    u64 seed;
    long max_attempts;
    const int64_t *ids;     // (B,) global instance ids, or null = instance0 + b
    long instance0;
    const int32_t *heights; // (B, S)
    int32_t *gt_blocks;     // (B, S*ns, 3)
    int32_t *gt_positions;  // (B, S*ns, 3), slabs stacked along z
    int32_t *attempts;      // (B, S) attempts used, -1 = cap reached
};

// one thread = one slab of one instance: BPP_Generator_3D(ns, [W, W, h]) until check_all_blocks_size accepts
__global__ void __launch_bounds__(TAP_BLOCK) k_ppsg_gt(PpsgGtArgs a)
{
    const long t = (long)blockIdx.x * TAP_BLOCK + threadIdx.x;
    if (t >= (long)a.B * a.S) return;
    const int b = (int)(t / a.S), s = (int)(t - (long)b * a.S);
    const long inst = a.ids ? (long)a.ids[b] : a.instance0 + b;
    const int ns = a.ns, h = a.heights[(size_t)b * a.S + s];
    int blk[PPSG_MAX_SLAB][3], pos[PPSG_MAX_SLAB][3];
    long vol[PPSG_MAX_SLAB];
    long used = 0;
    bool ok = false;
    for (long att = 0; att < a.max_attempts && !ok; ++att) {
        PpsgRng r = {ppsg_key(a.seed, (u64)(inst * a.S + s), (u64)a.gen, (u64)att), 0};
        for (int i = 0; i < ns; ++i) for (int k = 0; k < 3; ++k) { blk[i][k] = 0; pos[i][k] = 0; }
        blk[0][0] = a.W; blk[0][1] = a.W; blk[0][2] = h;                     // :249
        vol[0] = (long)a.W * a.W * h;
        for (int bi = 1; bi < ns; ++bi) {
            int c = 0;
            if (bi > 1) c = r.choice_weighted(vol, bi);                      // :256-267
            const long dims[3] = {blk[c][0], blk[c][1], blk[c][2]};
            const int axis = r.choice_weighted(dims, 3);                     // :276-277
            const int axis_max = blk[c][axis];
            const int mn = a.min_size, mx = min(axis_max, a.max_size);       // :281-282
            const int split = (mn >= mx) ? mn : (int)r.randint(mn, mx);      // :284-287 (mn > mx: the reference raises;
                                                                             //  unreachable for min_size = 1)
            for (int k = 0; k < 3; ++k) { blk[bi][k] = blk[c][k]; pos[bi][k] = pos[c][k]; }
            const int first = min(split, axis_max);                          // :290-295
            blk[c][axis] = first;
            blk[bi][axis] = axis_max - first;
            pos[bi][axis] += split;
            vol[c] = (long)blk[c][0] * blk[c][1] * blk[c][2];                // :298-299
            vol[bi] = (long)blk[bi][0] * blk[bi][1] * blk[bi][2];
        }
        ok = true;                                                           // :41-53
        for (int i = 0; i < ns; ++i)
            for (int k = 0; k < 3; ++k) ok &= blk[i][k] >= a.min_size && blk[i][k] < a.max_size;
        ++used;
    }
    int zoff = 0;
    for (int q = 0; q < s; ++q) zoff += a.heights[(size_t)b * a.S + q];
    for (int i = 0; i < ns; ++i) {
        const size_t o = (((size_t)b * a.S + s) * ns + i) * 3;
        a.gt_blocks[o] = blk[i][0]; a.gt_blocks[o + 1] = blk[i][1]; a.gt_blocks[o + 2] = blk[i][2];
        a.gt_positions[o] = pos[i][0]; a.gt_positions[o + 1] = pos[i][1]; a.gt_positions[o + 2] = pos[i][2] + zoff;
    }
    if (a.attempts) a.attempts[(size_t)b * a.S + s] = ok ? (int)min(used, (long)INT_MAX) : -1;
}

struct PpsgOrderArgs {
    int B, n, gen, trial;
    u64 seed;
    const int64_t *ids;
    long instance0;
    const int32_t *gt_blocks, *gt_positions; // (B, n, 3)
    int32_t *blocks_out;                     // (B, n, 3) layout order, rotated
};

// one thread = one instance: generate.py:86-105
template <int D>
__global__ void __launch_bounds__(TAP_BLOCK) k_ppsg_order(PpsgOrderArgs a)
{
    const int b = blockIdx.x * TAP_BLOCK + threadIdx.x;
    if (b >= a.B) return;
    const int n = a.n;
    const long inst = a.ids ? (long)a.ids[b] : a.instance0 + b;
    const int32_t *gb = a.gt_blocks + (size_t)b * n * D, *gp = a.gt_positions + (size_t)b * n * D;
    // on[j] = blocks that must leave before j.  3D: the packing has no holes, so calc_dependent_3D's "nearest
    // voxel below / above in each column" (generate.py:674-701) is the block whose bottom touches j's top with
    // an overlapping footprint.  2D: calc_dependent's movement rule (generate.py:575-647), every block that
    // starts higher and overlaps j's x range.
    u64 on[64];
    for (int j = 0; j < n; ++j) {
        u64 m = 0;
        if (D == 3) {
            const int jx = gp[j * 3], jy = gp[j * 3 + 1], jt = gp[j * 3 + 2] + gb[j * 3 + 2];
            for (int i = 0; i < n; ++i) {
                if (i == j || gp[i * 3 + 2] != jt) continue;
                const int ix = gp[i * 3], iy = gp[i * 3 + 1];
                if (ix < jx + gb[j * 3] && jx < ix + gb[i * 3] && iy < jy + gb[j * 3 + 1] && jy < iy + gb[i * 3 + 1]) m |= 1ull << i;
            }
        } else {
            const int jx = gp[j * 2], jz = gp[j * 2 + 1];
            for (int i = 0; i < n; ++i) {
                if (i == j) continue;
                const int ix = gp[i * 2];
                if (ix < jx + gb[j * 2] && jx < ix + gb[i * 2] && gp[i * 2 + 1] > jz) m |= 1ull << i;
            }
        }
        on[j] = m;
    }
    PpsgRng r = {ppsg_key(a.seed, (u64)inst, (u64)a.gen, (u64)(1000 + a.trial)), 0};
    u64 chosen = 0;
    int order[64], cnt = 0;
    // :89 (all deps are gone once every block is chosen).  A draw hits an unchosen candidate with probability
    // >= 1/n, so the guard only ever ends the loop on corrupt input (relations with a cycle); the rest is
    // then appended in index order instead of spinning on the GPU.
    u64 cand = 0;                                            // :91 rows of my_deps that sum to 0, ascending; the set only
    for (int j = 0; j < n; ++j) if (on[j] == 0) cand |= 1ull << j;   // changes when a block is chosen (:98)
    for (int guard = 0; cnt < n && guard < (1 << 20); ++guard) {
        const int nc = __popcll(cand);
        if (nc == 0) break;                                  // cannot happen: a packing always has a top block
        int k = (int)r.randint(0, nc);                       // :93 np.random.choice(candidate_idx)
        u64 c = cand;
        while (k--) c &= c - 1;
        const int idx = __ffsll((long long)c) - 1;
        if ((chosen >> idx) & 1ull) continue;                // :94-95
        chosen |= 1ull << idx;                               // :97-98
        order[cnt++] = idx;
        for (int j = 0; j < n; ++j) if ((on[j] & ~chosen) == 0) cand |= 1ull << j;
    }
    for (int j = 0; j < n && cnt < n; ++j) if (!((chosen >> j) & 1ull)) { chosen |= 1ull << j; order[cnt++] = j; }
    const int perms3[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    const int perms2[2][3] = {{0, 1, 0}, {1, 0, 0}};
    int32_t *out = a.blocks_out + (size_t)b * n * D;
    for (int i = 0; i < cnt; ++i) {                          // :100, :103-105
        const int32_t *g = gb + order[i] * D;
        const int p = (int)r.randint(0, D == 3 ? 6 : 2);
        for (int k = 0; k < D; ++k) out[i * D + k] = g[D == 3 ? perms3[p][k] : perms2[p][k]];
    }
}

// ---- 2D: generate.BPP_Generator_2D_easy (generate.py:392-484), what generate_blocks_with_GT calls for block_dim 2
struct PpsgGt2Args {
    int B, n, W, min_size, max_size, gen;
    u64 seed;
    long max_attempts;
    const int64_t *ids;
    long instance0;
    const int32_t *heights; // (B,) height of the perfect packing
    const double *gauss;    // (rows, stride): split table of the Gaussian branch, built by the caller with numpy
    int gstride, grows;
    int32_t *gt_blocks;     // (B, n, 2)
    int32_t *gt_positions;  // (B, n, 2)
    int32_t *attempts;      // (B,) attempts used, -1 = cap reached
};

// one thread = one instance: the generator repeated until check_all_blocks_size accepts (generate.py:66-75)
__global__ void __launch_bounds__(TAP_BLOCK) k_ppsg_gt2d(PpsgGt2Args a)
{
    const int b = blockIdx.x * TAP_BLOCK + threadIdx.x;
    if (b >= a.B) return;
    const long inst = a.ids ? (long)a.ids[b] : a.instance0 + b;
    const int n = a.n, H = a.heights[b], mn = a.min_size, mx = a.max_size;
    int blk[64][2], pos[64][2], ids[64];
    long vol[64], w[64];
    long used = 0;
    bool ok = false;
    for (long att = 0; att < a.max_attempts && !ok; ++att) {
        PpsgRng r = {ppsg_key(a.seed, (u64)inst, (u64)a.gen, (u64)att), 0};
        for (int i = 0; i < n; ++i) { blk[i][0] = blk[i][1] = 0; pos[i][0] = pos[i][1] = 0; }
        blk[0][0] = a.W; blk[0][1] = H;                                      // :410
        vol[0] = (long)a.W * H;
        bool raised = false;
        for (int bi = 1; bi < n && !raised; ++bi) {
            int c = 0;
            if (bi > 1) {                                                    // :417-436
                int k = 0;
                for (int id = 0; id < bi; ++id) if (blk[id][0] >= mx || blk[id][1] >= mx) ids[k++] = id;
                if (k == 0)
                    for (int id = 0; id < bi; ++id) if (blk[id][0] >= 2 * mn || blk[id][1] >= 2 * mn) ids[k++] = id;
                if (k == 0) { raised = true; break; }                        // the reference raises: unreachable for W >= 2
                for (int i = 0; i < k; ++i) w[i] = vol[ids[i]];
                c = ids[r.choice_weighted(w, k)];
            }
            const int X = blk[c][0], Z = blk[c][1], x = pos[c][0];
            int axis;
            if (X >= 2 * mx && X > Z) axis = 0;                              // :446
            else if (Z >= 2 * mx && Z > x) axis = 1;                         // :447 (sic: the POSITION x)
            else {
                const long dims[2] = {X, Z};
                axis = r.choice_weighted(dims, 2);                           // :449-450
                if (blk[c][axis] < 2 * mn) axis ^= 1;                        // :451-454
            }
            const int len = blk[c][axis];
            int split;
            if (len < 2 * mx - 1) {                                          // :458-461
                const int hi = min(len, mx) - mn + 1;
                if (mn >= hi) { raised = true; break; }                      // randint(low >= high) raises
                split = (int)r.randint(mn, hi);
            } else {                                                         // :463-469
                const int m = len - 2 * mn;
                if (len >= a.grows || m > a.gstride) { raised = true; break; }
                const double *cdf = a.gauss + (size_t)len * a.gstride;
                const double u = r.sample();
                int idx = 0;
                while (idx < m && cdf[idx] <= u) ++idx;
                if (idx >= m) idx = m - 1;
                split = mn + idx;
            }
            blk[bi][0] = blk[c][0]; blk[bi][1] = blk[c][1];                  // :472-476
            pos[bi][0] = pos[c][0]; pos[bi][1] = pos[c][1];
            blk[c][axis] = split;
            blk[bi][axis] = len - split;
            pos[bi][axis] += split;
            vol[c] = (long)blk[c][0] * blk[c][1];                            // :479-480
            vol[bi] = (long)blk[bi][0] * blk[bi][1];
        }
        ++used;
        if (raised) break;
        ok = true;                                                           // :41-53
        for (int i = 0; i < n; ++i) ok &= blk[i][0] >= mn && blk[i][0] < mx && blk[i][1] >= mn && blk[i][1] < mx;
    }
    for (int i = 0; i < n; ++i) {
        const size_t o = ((size_t)b * n + i) * 2;
        a.gt_blocks[o] = blk[i][0]; a.gt_blocks[o + 1] = blk[i][1];
        a.gt_positions[o] = pos[i][0]; a.gt_positions[o + 1] = pos[i][1];
    }
    if (a.attempts) a.attempts[b] = ok ? (int)min(used, (long)INT_MAX) : -1;
}

// generate.py:110-156 on the relations tap_rolling_init derived from the packed layout: every block stable,
// and -- last packed first -- each block has nothing on it and one free side per horizontal axis
__global__ void __launch_bounds__(TAP_BLOCK) k_ppsg_check(int B, int n, int input_simple, const unsigned long long *rel,
                                                          const uint8_t *stable, uint8_t *ok_out)
{
    const int b = blockIdx.x * TAP_BLOCK + threadIdx.x;
    if (b >= B) return;
    bool ok = true;
    for (int i = 0; i < n; ++i) ok &= stable[(size_t)b * n + i] != 0;        // :110
    const u64 *r = rel + (size_t)b * 5 * n;
    u64 left = n == 64 ? ~0ull : ((1ull << n) - 1ull);                        // blocks still in the container
    for (int s = n - 1; s >= 0 && ok; --s) {
        const u64 *q = r + n + 4 * s;                                       // tap_rolling_init's layout (tapenv.h)
        const u64 mv = r[s] & left, lf = q[0] & left, rt = q[1] & left, fw = q[2] & left, bw = q[3] & left;
        const bool x = fw && bw, y = lf && rt;
        if (mv == 0 && (input_simple || (!x && !y))) left &= ~(1ull << s);   // :142-149
        else ok = false;
    }
    ok_out[b] = ok;
}

extern "C" int tap_ppsg_gt(tap_ctx *ctx, int B, int S, int ns, int W, const int32_t *heights, int min_size,
                           int max_size, uint64_t seed, const int64_t *ids, int64_t instance0, int gen,
                           int64_t max_attempts, int32_t *gt_blocks_out, int32_t *gt_positions_out,
                           int32_t *attempts_out, void *stream)
{
    if (B < 0 || S < 1 || ns < 1 || ns > PPSG_MAX_SLAB || W < 1 || min_size < 1 || max_size <= min_size || max_attempts < 1)
        return tap_fail(ctx, TAP_E_INVALID, "bad ppsg_gt arguments (blocks per slab <= %d)", PPSG_MAX_SLAB);
    if (B == 0) return TAP_OK;
    if (!heights || !gt_blocks_out || !gt_positions_out) return tap_fail(ctx, TAP_E_INVALID, "null pointer");
    PpsgGtArgs a = {B, S, ns, W, min_size, max_size, gen, (u64)seed, (long)max_attempts, ids, (long)instance0,
                    heights, gt_blocks_out, gt_positions_out, attempts_out};
    const long threads = (long)B * S;
    hipLaunchKernelGGL(k_ppsg_gt, dim3((unsigned)((threads + TAP_BLOCK - 1) / TAP_BLOCK)), dim3(TAP_BLOCK), 0,
                       (hipStream_t)stream, a);
    TAP_LAUNCH_CHECK(ctx, "k_ppsg_gt");
    return TAP_OK;
}

extern "C" int tap_ppsg_order(tap_ctx *ctx, int B, int n, const int32_t *gt_blocks, const int32_t *gt_positions,
                              uint64_t seed, const int64_t *ids, int64_t instance0, int gen, int trial,
                              int32_t *blocks_out, void *stream)
{
    if (B < 0 || n < 1 || n > 64) return tap_fail(ctx, TAP_E_INVALID, "bad ppsg_order arguments (n <= 64)");
    if (B == 0) return TAP_OK;
    if (!gt_blocks || !gt_positions || !blocks_out) return tap_fail(ctx, TAP_E_INVALID, "null pointer");
    PpsgOrderArgs a = {B, n, gen, trial, (u64)seed, ids, (long)instance0, gt_blocks, gt_positions, blocks_out};
    hipLaunchKernelGGL(k_ppsg_order<3>, dim3((B + TAP_BLOCK - 1) / TAP_BLOCK), dim3(TAP_BLOCK), 0, (hipStream_t)stream, a);
    TAP_LAUNCH_CHECK(ctx, "k_ppsg_order");
    return TAP_OK;
}

extern "C" int tap_ppsg_order2d(tap_ctx *ctx, int B, int n, const int32_t *gt_blocks, const int32_t *gt_positions,
                                uint64_t seed, const int64_t *ids, int64_t instance0, int gen, int trial,
                                int32_t *blocks_out, void *stream)
{
    if (B < 0 || n < 1 || n > 64) return tap_fail(ctx, TAP_E_INVALID, "bad ppsg_order2d arguments (n <= 64)");
    if (B == 0) return TAP_OK;
    if (!gt_blocks || !gt_positions || !blocks_out) return tap_fail(ctx, TAP_E_INVALID, "null pointer");
    PpsgOrderArgs a = {B, n, gen, trial, (u64)seed, ids, (long)instance0, gt_blocks, gt_positions, blocks_out};
    hipLaunchKernelGGL(k_ppsg_order<2>, dim3((B + TAP_BLOCK - 1) / TAP_BLOCK), dim3(TAP_BLOCK), 0, (hipStream_t)stream, a);
    TAP_LAUNCH_CHECK(ctx, "k_ppsg_order2d");
    return TAP_OK;
}

extern "C" int tap_ppsg_gt2d(tap_ctx *ctx, int B, int n, int W, const int32_t *heights, int min_size, int max_size,
                             const double *gauss, int gauss_stride, int gauss_rows, uint64_t seed, const int64_t *ids,
                             int64_t instance0, int gen, int64_t max_attempts, int32_t *gt_blocks_out,
                             int32_t *gt_positions_out, int32_t *attempts_out, void *stream)
{
    if (B < 0 || n < 1 || n > 64 || W < 1 || min_size < 1 || max_size <= min_size || max_attempts < 1 || gauss_stride < 1 ||
        gauss_rows < 1)
        return tap_fail(ctx, TAP_E_INVALID, "bad ppsg_gt2d arguments (n <= 64)");
    if (B == 0) return TAP_OK;
    if (!heights || !gauss || !gt_blocks_out || !gt_positions_out) return tap_fail(ctx, TAP_E_INVALID, "null pointer");
    PpsgGt2Args a = {B, n, W, min_size, max_size, gen, (u64)seed, (long)max_attempts, ids, (long)instance0, heights,
                     gauss, gauss_stride, gauss_rows, gt_blocks_out, gt_positions_out, attempts_out};
    hipLaunchKernelGGL(k_ppsg_gt2d, dim3((B + TAP_BLOCK - 1) / TAP_BLOCK), dim3(TAP_BLOCK), 0, (hipStream_t)stream, a);
    TAP_LAUNCH_CHECK(ctx, "k_ppsg_gt2d");
    return TAP_OK;
}

extern "C" int tap_ppsg_check(tap_ctx *ctx, int B, int n, int input_simple, const uint64_t *rel, const uint8_t *stable,
                              uint8_t *ok_out, void *stream)
{
    if (B < 0 || n < 1 || n > 64) return tap_fail(ctx, TAP_E_INVALID, "bad ppsg_check arguments (n <= 64)");
    if (B == 0) return TAP_OK;
    if (!rel || !stable || !ok_out) return tap_fail(ctx, TAP_E_INVALID, "null pointer");
    hipLaunchKernelGGL(k_ppsg_check, dim3((B + TAP_BLOCK - 1) / TAP_BLOCK), dim3(TAP_BLOCK), 0, (hipStream_t)stream, B, n,
                       input_simple, reinterpret_cast<const unsigned long long *>(rel), stable, ok_out);
    TAP_LAUNCH_CHECK(ctx, "k_ppsg_check");
    return TAP_OK;
}
