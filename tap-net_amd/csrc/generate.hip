// generate.hip -- instance generation on the device: the part of
// generate.generate_blocks (generate.py:773-971) after the block sizes have been drawn --
// precedence extraction (generate.calc_dependent, generate.py:575-771) from the packed initial
// container, laid out directly as the tensors pack.PACKDataset builds (pack.py:101-195,
// input_type 'bot', allow_rot=True).  The packing itself is tap_pack_blocks (env.hip) with the
// 'C+P+S-lb-hard' descriptor, generate.py:908.  gfx950 only.
//
// The reference scans the voxel grid around every block.  Blocks never intersect, so every scan is
// a pairwise box predicate on (position, size):
//   2D move    a blocks b  <=>  x-ranges overlap and z_a > z_b                  (generate.py:602-613)
//   3D move    a blocks b  <=>  in some column of b's footprint, a is the nearest block above
//                                                                              (generate.py:674-701)
//   sides      a blocks b's left/right/forward/backward  <=>  a owns a voxel in the probe region next
//              to b from z_mid up: a's footprint meets the probe cells and top_a > z_mid
//              (generate.py:615-641, 704-746); a block touching the wall blocks itself.
// Mapping: G = 16/32/64 lanes per instance, lane = block b; the lane builds the bitmask of blocks
// that block b (one column of each dependency matrix) and writes its columns of `dynamic`, so for
// every row the lanes of a group store consecutive floats.
#include "tap_common.h"
#include "tap_place.h"

struct PrecArgs {
    int B, D, n, W, L, H, arm;
    const int32_t *blocks;    // (B, n, D)
    const int32_t *positions; // (B, n, D)
    float *static_out;        // (B, 1+D, n*R)
    float *dynamic_out;       // (B, 3n, n*R)
};

__device__ __forceinline__ bool ranges_meet(int a0, int a1, int b0, int b1) { return a0 < b1 && b0 < a1; }

template <int D, int G>
__global__ void __launch_bounds__(TAP_BLOCK) k_precedence(PrecArgs a)
{
    __shared__ int s_blk[TAP_BLOCK / G][64 * 3];
    __shared__ int s_pos[TAP_BLOCK / G][64 * 3];
    const int tid = threadIdx.x, grp = tid / G, b = tid % G;
    const int inst = blockIdx.x * (TAP_BLOCK / G) + grp;
    const int n = a.n;
    if (inst < a.B)
        for (int k = b; k < n * D; k += G) {
            s_blk[grp][k] = a.blocks[(size_t)inst * n * D + k];
            s_pos[grp][k] = a.positions[(size_t)inst * n * D + k];
        }
    tap_wave_lds_sync();
    if (inst >= a.B || b >= n) return;
    const int *blk = s_blk[grp], *pos = s_pos[grp];
#define BX(i) blk[(i) * D]
#define BY(i) (D == 3 ? blk[(i) * D + 1] : 1)
#define BZ(i) blk[(i) * D + D - 1]
#define PX(i) pos[(i) * D]
#define PY(i) (D == 3 ? pos[(i) * D + 1] : 0)
#define PZ(i) pos[(i) * D + D - 1]
    const int x = PX(b), y = PY(b), z = PZ(b), bx = BX(b), by = BY(b), bz = BZ(b);
    const int top = z + bz;
    const int z_mid = z + (bz - 1) / 2;                               // generate.py:615, 704
    u64 move = 0, left = 0, right = 0, fwd = 0, bwd = 0;

    if (D == 2) {
        for (int o = 0; o < n; ++o) {
            if (o == b) continue;
            const int ox = PX(o), oz = PZ(o), obx = BX(o), otop = oz + BZ(o);
            if (ranges_meet(ox, ox + obx, x, x + bx) && oz > z) move |= 1ull << o;          // :602-613
            if (x >= a.arm && ranges_meet(ox, ox + obx, x - a.arm, x) && otop > z_mid) left |= 1ull << o;   // :627-631
            if (x + bx <= a.W - a.arm && ranges_meet(ox, ox + obx, x + bx, x + bx + a.arm) && otop > z_mid)
                right |= 1ull << o;                                                         // :637-641
        }
        if (x < a.arm) left |= 1ull << b;                                                   // :623-625
        if (x + bx > a.W - a.arm) right |= 1ull << b;                                       // :634-635
    } else {
        // nearest block above, per footprint column (generate.py:689-701 and its mirror :674-687)
        for (int i = 0; i < bx; ++i)
            for (int j = 0; j < by; ++j) {
                const int cx = x + i, cy = y + j;
                int best = -1, bestz = INT_MAX;
                for (int o = 0; o < n; ++o) {
                    if (o == b) continue;
                    const int ox = PX(o), oy = PY(o), oz = PZ(o);
                    if (cx >= ox && cx < ox + BX(o) && cy >= oy && cy < oy + BY(o) && oz >= top && oz < bestz) {
                        bestz = oz; best = o;
                    }
                }
                if (best >= 0) move |= 1ull << best;
            }
        const int ymid = y + (by - 1) / 2, xmid = x + (bx - 1) / 2;  // :705-713: a single probe cell
        for (int o = 0; o < n; ++o) {
            if (o == b) continue;
            const int ox = PX(o), oy = PY(o), otop = PZ(o) + BZ(o);
            if (otop <= z_mid) continue;
            const bool in_y = ymid >= oy && ymid < oy + BY(o), in_x = xmid >= ox && xmid < ox + BX(o);
            if (x > 0 && in_y && x - 1 >= ox && x - 1 < ox + BX(o)) left |= 1ull << o;        // :720-722
            if (x + bx < a.W && in_y && x + bx >= ox && x + bx < ox + BX(o)) right |= 1ull << o; // :728-730
            if (y > 0 && in_x && y - 1 >= oy && y - 1 < oy + BY(o)) fwd |= 1ull << o;         // :736-738
            if (y + by < a.L && in_x && y + by >= oy && y + by < oy + BY(o)) bwd |= 1ull << o; // :744-746
        }
        if (x == 0) left |= 1ull << b;                                                      // :716-718
        if (x + bx == a.W) right |= 1ull << b;                                              // :725-726
        if (y == 0) fwd |= 1ull << b;                                                       // :733-734
        if (y + by == a.L) bwd |= 1ull << b;                                                // :741-742
    }

    // PACKDataset layout (pack.py:101-195): column r*n + b of every row
    constexpr int R = D == 2 ? 2 : 6;
    const int perm2[2][3] = {{0, 1, 0}, {1, 0, 0}};
    const int perm3[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    const int nR = n * R;
    float *st = a.static_out + (size_t)inst * (1 + D) * nR;
    float *dy = a.dynamic_out + (size_t)inst * 3 * n * nR;
    for (int r = 0; r < R; ++r) {
        const int *p = D == 2 ? perm2[r] : perm3[r];
        const int col = r * n + b;
        st[col] = (float)b;                                                                 // pack.py:144-147
        for (int k = 0; k < D; ++k) st[(size_t)(1 + k) * nR + col] = (float)blk[b * D + p[k]]; // generate.py:951,960
        u64 small = 0, large = 0;                                                           // generate.py:941-959
        if (p[D - 1] == 0) { small = left; large = right; }
        else if (D == 3 && p[D - 1] == 1) { small = fwd; large = bwd; }
        for (int o = 0; o < n; ++o) {
            dy[(size_t)o * nR + col] = (float)((move >> o) & 1);
            dy[(size_t)(n + o) * nR + col] = (float)((small >> o) & 1);
            dy[(size_t)(2 * n + o) * nR + col] = (float)((large >> o) & 1);
        }
    }
#undef BX
#undef BY
#undef BZ
#undef PX
#undef PY
#undef PZ
}

template <int D, int G> static int launch_prec(tap_ctx *ctx, const PrecArgs &a, hipStream_t st)
{
    const int ipb = TAP_BLOCK / G, grid = (a.B + ipb - 1) / ipb;
    if (grid == 0) return TAP_OK;
    hipLaunchKernelGGL((k_precedence<D, G>), dim3(grid), dim3(TAP_BLOCK), 0, st, a);
    TAP_LAUNCH_CHECK(ctx, "k_precedence");
    return TAP_OK;
}

extern "C" int tap_precedence(tap_ctx *ctx, int B, int D, int n, const int32_t *container_size,
                              int arm_size, const int32_t *blocks, const int32_t *positions,
                              float *static_out, float *dynamic_out, void *stream)
{
    if ((D != 2 && D != 3) || B < 0 || n < 1 || n > 64 || !container_size || arm_size < 1 || !blocks ||
        !positions || !static_out || !dynamic_out)
        return tap_fail(ctx, TAP_E_INVALID, "bad precedence arguments (n must be <= 64)");
    PrecArgs a = {B, D, n, container_size[0], D == 3 ? container_size[1] : 1, container_size[D - 1],
                  arm_size, blocks, positions, static_out, dynamic_out};
    hipStream_t st = (hipStream_t)stream;
    if (D == 2) {
        if (n <= 16) return launch_prec<2, 16>(ctx, a, st);
        if (n <= 32) return launch_prec<2, 32>(ctx, a, st);
        return launch_prec<2, 64>(ctx, a, st);
    }
    if (n <= 16) return launch_prec<3, 16>(ctx, a, st);
    if (n <= 32) return launch_prec<3, 32>(ctx, a, st);
    return launch_prec<3, 64>(ctx, a, st);
}
