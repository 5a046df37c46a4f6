// macs.hip -- MACS / MUL 2D placement: tools.calc_one_position_mcs_2d (tools.py:2456-2749),
// re-stated on the height-map plus the placement history.  gfx950 only.
//
// The reference keeps, per container, a voxel grid and one free-interval list per level, builds a
// list of "empty maximal spaces" (EMS) and walks both bottom corners of every EMS sequentially,
// sliding the block until it settles, with a `visited` set shared by all walks.
//
// Mapping: G = 8 (W <= 8) or 16 lanes per env, lane = container column, 256-thread workgroups
// (256/G envs).  What makes the walk parallel:
//   * whether a block settles at position (x, Z) -- supported, free, and stable when the reward is
//     'hard' -- is a property of (x, Z) alone.  A walk therefore stops at the first position in
//     its direction that is `good` and has not been taken by an earlier walk: positions an earlier
//     walk examined and rejected would be rejected again, positions it accepted are exactly the
//     ones it settled on.  So per EMS every lane tests its own column once, one wave ballot gives
//     the good-mask of the level, and each walk is a find-first-set on
//     good & ~taken[Z] & range.  `taken` is a per-level bitmask in LDS.
//   * the EMS list is built by all lanes of the group redundantly in lock-step (every lane writes
//     the same words to the group's LDS slice and reads back only what it wrote itself, so no
//     barrier is needed anywhere); the block-top de-duplication scan is strided over the lanes.
//   * voxel (c, z) != 0 <=> z < hm[c]; level_free_space[z] == maximal runs of columns with
//     hm[c] <= z, so only z = 0 and z in {hm[c]} can open new level-EMS.
//   * the usable-space tie-break of a candidate map hm' is sum_{h < max_h} maxrun_h(hm') =
//     base(hm') + (max_h - max(hm')) (W - 1) with max_h common to all tied candidates, so ties are
//     ordered by base(hm') - max(hm') (W - 1) and the selection streams; base() is evaluated with
//     one lane per threshold column.
// Checked against the reference's own traces (tests/golden/macs2d.npz) and the voxel-level oracle.
#include "tap_common.h"
#include "tap_place.h"

constexpr int MACS_EMS_CAP = 128; // packed EMS entries per env
constexpr int MACS_MAX_H = 256;

__host__ __device__ constexpr int macs_group_words(int G, int H, int n_max)
{
    return G + MACS_EMS_CAP + (H + 1) / 2 + 4 * n_max; // hm | ems | taken (uint16 per level) | history
}

template <int G> __device__ __forceinline__ int group_sum(int v)
{
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, G);
    return v;
}

template <int G>
__global__ void __launch_bounds__(TAP_BLOCK) k_macs2d_step(StepArgs a)
{
    extern __shared__ int lds[];
    const int tid = threadIdx.x, cell = tid % G;
    const int env = blockIdx.x * (TAP_BLOCK / G) + tid / G;
    const int B = a.d.B, W = a.d.W, H = a.d.H;
    const bool ev = env < B, incell = cell < W;
    const int gl0 = (tid & 63) - cell;
    int *g_hm = lds + (tid / G) * macs_group_words(G, H, a.d.n_max);
    int *g_ems = g_hm + G;
    unsigned short *g_taken = reinterpret_cast<unsigned short *>(g_ems + MACS_EMS_CAP);
    int *g_hist = g_ems + MACS_EMS_CAP + (H + 1) / 2; // (x, z, bx, bz) of every earlier step
#define HM(c) g_hm[c]

    int hm = (ev && incell) ? a.v.hm[(size_t)env * W + cell] : 0;
    const int cv = (ev && cell < 4) ? a.v.cnt[(size_t)env * 4 + cell] : 0;
    Counters cnt = {__shfl(cv, gl0), __shfl(cv, gl0 + 1), __shfl(cv, gl0 + 2), __shfl(cv, gl0 + 3)};
    int bx = 1, bz = 1;
    bool act = ev;
    if (ev) {
        if (a.static_) {
            const long p = (long)a.ptr[env];
            bx = (int)a.static_[((size_t)env * a.static_rows + 1) * a.nR + p];
            bz = (int)a.static_[((size_t)env * a.static_rows + 2) * a.nR + p];
        } else if (a.blocks_dtype == TAP_DT_F32) {
            bx = (int)((const float *)a.blocks)[(size_t)env * 2];
            bz = (int)((const float *)a.blocks)[(size_t)env * 2 + 1];
        } else {
            bx = ((const int32_t *)a.blocks)[(size_t)env * 2];
            bz = ((const int32_t *)a.blocks)[(size_t)env * 2 + 1];
        }
        if (a.active) act = a.active[env] != 0;
    }
    int err = 0;
    bool do_step = act;
    if (act && cnt.count >= a.d.n_max) { err |= 2; do_step = false; }
    if (act && (bx < 1 || bz < 1)) { err |= 4; do_step = false; }

    g_hm[cell] = hm;
    for (int i = cell; i < H; i += G) g_taken[i] = 0;
    if (ev) // one round trip for the whole placement history instead of one per earlier block
        for (int k = cell; k < cnt.count * 4 && k < a.d.n_max * 4; k += G) {
            const int i = k >> 2, f = k & 3;
            g_hist[k] = (f < 2 ? a.v.pos : a.v.blk)[(size_t)(i * 2 + (f & 1)) * B + env];
        }
    tap_wave_lds_sync();
    const int gmax = group_max<G>(incell ? hm : 0);

    int px = 0, pz = 0, pst = 0, placed = 0, emp_w = 0;
    const int step = cnt.count;
    if (do_step) { // group-uniform
        const int hard = a.d.flags & TAP_F_HARD;
        const int vol = bx * bz;
        // the block history later steps read (tools.py:2531-2533), failures too
        if (cell == 0) {
            a.v.blk[(size_t)(step * 2) * B + env] = bx;
            a.v.blk[(size_t)(step * 2 + 1) * B + env] = bz;
        }

        // ---- EMS list (all lanes in lock-step, identical writes) ------------------------------
        int n_ems = 0;
#define EMS_PUSH(x1, z, x2)                                                                  \
    do {                                                                                     \
        if (n_ems < MACS_EMS_CAP) g_ems[n_ems++] = ((x1) & 0xff) | (((x2) & 0xff) << 8) | ((z) << 16); \
        else err |= 16;                                                                      \
    } while (0)
        // (a) per-level free runs (tools.py:2517-2529)
        for (int z = 0;;) {
            if (z + bz > H) break;                                                // :2519
            int c = 0;
            while (c < W) {
                if (HM(c) > z) { ++c; continue; }
                const int x1 = c;
                bool opened = false; // a column of the run has hm == z: the run is new at level z
                while (c < W && HM(c) <= z) { opened |= HM(c) == z; ++c; }
                const int x2 = c - 1;
                if (x1 + bx > W) break;                                           // :2525
                if (z > 0 && !opened) continue;                                   // :2526-2528 same run below
                EMS_PUSH(x1, z, x2);                                              // :2529
            }
            const int nz = group_min<G>((incell && hm > z) ? hm : INT_MAX);       // :2520 next level that differs
            if (nz == INT_MAX) break;
            z = nz;
        }
        // (b) tops of the blocks placed so far (tools.py:2531-2555); failed steps sit at (0, 0)
        for (int i = 0; i < step; ++i) {
            const int x = g_hist[i * 4], z = g_hist[i * 4 + 1], xx = g_hist[i * 4 + 2], zz = g_hist[i * 4 + 3];
            const int tz = z + zz;
            if (!(tz < H)) continue;                                              // :2535
            // :2537 all columns under the block's top are free at level tz (slice clips at W)
            const bool full = group_or<G>((incell && cell >= x && cell < x + xx && hm > tz) ? 1 : 0) == 0;
            if (full) {
                const int want = (x & 0xff) | (((x + xx - 1) & 0xff) << 8) | (tz << 16);
                int dup = 0;                                                      // :2538
                for (int k = cell; k < n_ems; k += G) dup |= g_ems[k] == want;
                if (!group_or<G>(dup)) EMS_PUSH(x, tz, x + xx - 1);
            } else {
                if (x + xx - 1 >= W) { err |= 8; continue; }                      // reference: IndexError :2550
                if (HM(x) <= tz && x > 0 && HM(x - 1) <= tz) {                    // :2543-2548 left part
                    int x2 = x;
                    for (;;) {
                        if (x2 == W - 1 || HM(x2 + 1) > tz) break;
                        if (x2 == x + xx - 1) break;
                        ++x2;
                    }
                    EMS_PUSH(x, tz, x2);
                }
                if (HM(x + xx - 1) <= tz && x + xx < W && HM(x + xx) <= tz) {     // :2550-2555 right part
                    int x1 = x + xx - 1;
                    for (;;) {
                        if (x1 == 0 || HM(x1 - 1) > tz) break;
                        if (x1 == x) break;
                        --x1;
                    }
                    EMS_PUSH(x1, tz, x + xx - 1);
                }
            }
        }

        // ---- both corners of every EMS (tools.py:2680-2700), streaming selection (:2708-2736) ----
        const int X = W - bx + 1;
        const unsigned gmask = (1u << G) - 1u;
        double best = -1.0, Sv0 = 0.0, Sv1 = 0.0;
        if (a.d.flags & TAP_F_USE_S) {                                            // :2602-2604, both outcomes
            Sv0 = (double)cnt.nstable / (double)(cnt.count + 1);
            Sv1 = (double)(cnt.nstable + 1) / (double)(cnt.count + 1);
        }
        const int valid2 = cnt.valid + vol;
        int best_adj = INT_MIN;
        for (int e = 0; e < n_ems; ++e) {
            const int pk = g_ems[e];
            const int X1 = pk & 0xff, X2 = (pk >> 8) & 0xff, Z = pk >> 16;
            // every lane tests its own column as the block's left edge at level Z (:2571-2588)
            int mx = -1, sum = 0, stab = 0;
            bool good = false;
            if (incell && cell + bx <= W) {
                u64 eq = 0;
                for (int i = 0; i < bx; ++i) {
                    const int h = HM(cell + i);
                    sum += h;
                    if (h > mx) { mx = h; eq = 1ull << i; }
                    else if (h == mx) eq |= 1ull << i;
                }
                const bool supported = !(Z > 0 && mx < Z);                        // :2574
                const bool free_ = mx <= Z;                                       // :2576
                stab = (Z == 0) ? 1 : tap_stable2d(bx, eq);                       // :2577-2585
                good = supported && free_ && (stab || !hard);                     // :2580-2581
            }
            const unsigned gm = (unsigned)((__ballot(good) >> gl0) & gmask);
            unsigned tk = g_taken[Z];
            double Cv = -1.0; // compactness of this level, computed on first use
            for (int side = 0; side < 2; ++side) {
                unsigned m;
                if (side == 0) {                                                  // :2686 left corner, slide right
                    if (!(X1 < X)) continue;
                    m = gm & ~tk & ~((1u << X1) - 1u);
                } else {                                                          // :2694 right corner, slide left
                    const int hi = X2 - bx + 1;
                    if (hi < 0) continue;
                    if (hi + bx > W) { err |= 8; continue; }
                    m = gm & ~tk & ((2u << hi) - 1u);
                }
                if (!m) continue;
                const int xs = side == 0 ? __ffs((int)m) - 1 : 31 - __clz((int)m);
                tk |= 1u << xs;
                const int sstab = __shfl(stab, gl0 + xs), ssum = __shfl(sum, gl0 + xs);
                // calc_C_P_S (:2590-2604)
                const int top = Z + bz;
                const int mtrue = max(gmax, top);               // true max of the candidate map
                const int emp = cnt.empty + bx * Z - ssum;                        // :2598-2599
                double r = 0.0;
                if (!(a.d.flags & TAP_F_MCS_ZERO)) {                              // :2709-2712
                    if (Cv < 0.0) {
                        int height = mtrue;
                        if (Z + bx > height) height = Z + bz;                     // :2594 (sic block_x)
                        Cv = (double)valid2 / (double)((long long)height * W);
                    }
                    const double P = (a.d.flags & TAP_F_USE_P) ? (double)valid2 / (double)(emp + valid2) : 0.0;
                    r = (Cv + P) + (sstab ? Sv1 : Sv0);
                }
                const bool tie = placed && r == best && (a.d.flags & TAP_F_MCS_TIE);
                int adj = 0;
                if ((a.d.flags & TAP_F_MCS_TIE) && (!placed || r > best || tie)) {
                    // base = sum_{h < m} longest free run (length - 1) at level h (:2667-2678),
                    // piecewise between the distinct heights; lane j owns threshold column j
                    const int hcj = (cell >= xs && cell < xs + bx) ? top : hm;
                    bool first = true;
                    int next = mtrue, best_run = 0, run = -1;
                    for (int k = 0; k < W; ++k) {
                        const int hk = __shfl(hcj, gl0 + k);
                        if (hk == hcj && k < cell) first = false;
                        if (hk > hcj) next = min(next, hk);
                        if (hk <= hcj) { ++run; best_run = max(best_run, run); } else run = -1;
                    }
                    const int contrib = (incell && first && hcj < mtrue) ? (next - hcj) * best_run : 0;
                    adj = group_sum<G>(contrib) - mtrue * (W - 1);
                }
                // first settled maximum; ties resolved by the usable-space score when enabled
                if (!placed || r > best || (tie && adj > best_adj)) {
                    placed = 1; best = r; best_adj = adj; px = xs; pz = Z; pst = sstab; emp_w = emp;
                }
            }
            g_taken[Z] = (unsigned short)tk; // every lane stores the same value and reads back its own
        }

        // ---- commit (:2738-2747) ----------------------------------------------------------------
        if (placed) {
            if (incell && cell >= px && cell < px + bx) hm = pz + bz;
            cnt.valid += vol;
            cnt.empty = emp_w;
            cnt.nstable += pst;
            if (pz + bz > H) err |= 1;
        } else {
            px = pz = pst = 0;
        }
        cnt.count += 1;
    }
    err = group_or<G>(err);

    tap_wave_lds_sync();
    g_hm[cell] = hm;
    tap_wave_lds_sync();
    if (ev) {
        if (incell) a.v.hm[(size_t)env * W + cell] = hm;
        if (a.feature_out)
            tap_write_feature<2, G>(a.d.feature, W, 1, g_hm, cell, hm, a.feature_out + (size_t)env * a.flen);
        if (cell == 0) {
            if (do_step) {
                reinterpret_cast<int4 *>(a.v.cnt)[env] = make_int4(cnt.valid, cnt.empty, cnt.nstable, cnt.count);
                a.v.pos[(size_t)(step * 2) * B + env] = px;
                a.v.pos[(size_t)(step * 2 + 1) * B + env] = pz;
                a.v.stable[(size_t)step * B + env] = (uint8_t)pst;
            }
            if (err) a.v.err[env] |= err;
        }
    } else if (a.d.feature == TAP_FEAT_ZERO) {
        (void)group_min<G>(INT_MAX);
    }
#undef HM
#undef EMS_PUSH
}

template <int G> static int launch_macs(tap_ctx *ctx, const StepArgs &a, hipStream_t st)
{
    const tap_env_desc &d = a.d;
    const int epb = TAP_BLOCK / G, grid = (d.B + epb - 1) / epb;
    if (grid == 0) return TAP_OK;
    const size_t lds = (size_t)epb * macs_group_words(G, d.H, d.n_max) * sizeof(int);
    if (lds > 64 * 1024) return tap_fail(ctx, TAP_E_UNSUPPORTED, "MACS: H=%d blocks_num=%d need %zu bytes of LDS per workgroup", d.H, d.n_max, lds);
    hipLaunchKernelGGL(k_macs2d_step<G>, dim3(grid), dim3(TAP_BLOCK), lds, st, a);
    TAP_LAUNCH_CHECK(ctx, "k_macs2d_step");
    return TAP_OK;
}

int tap_macs2d_step(tap_ctx *ctx, const StepArgs &a, hipStream_t st)
{
    const tap_env_desc &d = a.d;
    if (d.D != 2) return tap_fail(ctx, TAP_E_UNSUPPORTED, "MACS/MUL is implemented for 2D only");
    if (d.W > 16 || d.H > MACS_MAX_H)
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "MACS supports W <= 16 and H <= %d", MACS_MAX_H);
    if ((d.W + 1) * ((d.W + 1) / 2) + 2 * d.n_max > MACS_EMS_CAP)
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "MACS: blocks_num %d too large for the EMS list", d.n_max);
    return d.W <= 8 ? launch_macs<8>(ctx, a, st) : launch_macs<16>(ctx, a, st);
}
