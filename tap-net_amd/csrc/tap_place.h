// tap_place.h -- device code: one LB_GREEDY placement for one container, G lanes per container.
//
// Mapping (CDNA4, wave64): a container ("env") owns a group of G = 8/16/32/64 consecutive lanes
// of one wavefront; lane `cell` of the group owns height-map cell (x, y) = (cell / L, cell % L).
// A 256-thread workgroup therefore steps 256/G envs and a launch of B envs has B*G/256 >= 256
// workgroups at the BASELINE batch sizes.  Every left-bottom corner candidate is evaluated by the
// lane that owns its cell, in parallel; the reference's "first maximum in (z, y, class, x) order"
// argmax (tools.py:2081, 2161-2162, 2250, 2262) becomes a log2(G)-step butterfly over
// (ratio desc, key asc).  The height-map slice of the group is staged in LDS so that footprint
// scans are LDS reads, not global reads.
//
// Only the height-map and four counters are carried (the reference's voxel
// grid is redundant); the oracle keeps the voxels, and the parity tests compare the two.
#pragma once

#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>

#include "tapenv.h"

typedef unsigned long long u64;

struct PlaceCfg {
    int W, L, H, flags;
    const uint32_t *lut; // is_stable look-up table for footprints <= 4x4 (env.hip), may be null
};

struct Counters {
    int valid, empty, nstable, count;
};

struct Placement {
    int placed; // group-uniform
    int x, y, z, stab;
};

// ---- cross-lane helpers (all lanes of the group must be active) ----------------------------
// A group-wide combine never goes through the LDS crossbar (ds_bpermute, what __shfl_xor compiles to: an LDS
// round trip per step): inside a 16-lane row the partner's value arrives by a DPP-modified move (quad
// permutes, then the half-row and row mirrors -- each step merges two disjoint lane sets, which is all a
// commutative combine needs), and once a row is uniform, rows are combined through v_readlane of their first
// lanes.  group_butterfly<G>(lane, f) calls f(get) once per step; get(v) is the partner's v (an int).
template <int CTRL> __device__ __forceinline__ int tap_dpp(int v)
{
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false);
}
template <int G, typename F> __device__ __forceinline__ void group_butterfly(int lane /* 0..63 in the wave */, F f)
{
    static_assert(G == 8 || G == 16 || G == 32 || G == 64, "lane groups are 8, 16, 32 or 64 wide");
    f([](int v) { return tap_dpp<0xB1>(v); });                       // quad_perm [1,0,3,2]
    f([](int v) { return tap_dpp<0x4E>(v); });                       // quad_perm [2,3,0,1]
    f([](int v) { return tap_dpp<0x141>(v); });                      // row_half_mirror: the other quad of the 8
    if (G >= 16) f([](int v) { return tap_dpp<0x140>(v); });         // row_mirror: the other half of the row
    if (G >= 32)                                                      // the neighbouring row (rows are uniform by now)
        f([lane](int v) {
            const int r0 = __builtin_amdgcn_readlane(v, 0), r1 = __builtin_amdgcn_readlane(v, 16),
                      r2 = __builtin_amdgcn_readlane(v, 32), r3 = __builtin_amdgcn_readlane(v, 48);
            return lane < 32 ? (lane < 16 ? r1 : r0) : (lane < 48 ? r3 : r2);
        });
    if (G == 64)                                                      // the other half of the wave
        f([lane](int v) {
            const int r0 = __builtin_amdgcn_readlane(v, 0), r2 = __builtin_amdgcn_readlane(v, 32);
            return lane < 32 ? r2 : r0;
        });
}
template <int G> __device__ __forceinline__ int group_max(int v)
{
    group_butterfly<G>((int)(threadIdx.x & 63), [&](auto get) { v = max(v, get(v)); });
    return v;
}
template <int G> __device__ __forceinline__ int group_min(int v)
{
    group_butterfly<G>((int)(threadIdx.x & 63), [&](auto get) { v = min(v, get(v)); });
    return v;
}
template <int G> __device__ __forceinline__ int group_or(int v)
{
    group_butterfly<G>((int)(threadIdx.x & 63), [&](auto get) { v |= get(v); });
    return v;
}
template <int G> __device__ __forceinline__ int group_sum(int v)
{
    group_butterfly<G>((int)(threadIdx.x & 63), [&](auto get) { v += get(v); });
    return v;
}
template <int G> __device__ __forceinline__ double group_fmax(double v)
{
    group_butterfly<G>((int)(threadIdx.x & 63), [&](auto get) {
        v = fmax(v, __hiloint2double(get(__double2hiint(v)), get(__double2loint(v))));
    });
    return v;
}

// ---- stability predicates on a support mask ------------------------------------------------
// `eq` marks the footprint cells whose column top equals the placement level z (the cells that
// carry the block); every other footprint cell is lower.

// tools.is_stable_2d (tools.py:839-868): bit i = column x+i.  The centre x + bx/2 must lie
// strictly inside (first supported, last supported + 1)  <=>  2*lead < bx and 2*trail < bx.
__device__ __forceinline__ int tap_stable2d(int bx, u64 eq)
{
    const int lead = __ffsll((long long)eq) - 1;
    const int trail = bx - 1 - (63 - __clzll((long long)eq));
    return (2 * lead < bx) && (2 * trail < bx);
}

// tools.py:736-744 / 755-762, doubled coordinates
__device__ __forceinline__ int tap_line_rule(int cx, int cy, int p0x, int p0y, int p1x, int p1y)
{
    const int a = cx - p0x, b = cy - p0y, c = cx - p1x, d = cy - p1y;
    if (b == 0 || d == 0) return (b == d) && ((a < 0) != (c < 0));
    return (a * d == c * b) && ((a < 0) != (c < 0)) && ((b < 0) != (d < 0));
}

// tools.is_stable (tools.py:710-765) for z > 0: bit (i*8 + j) = footprint cell (i, j) supports.
// The reference builds scipy's convex hull and asks matplotlib's Path.contains_point (Agg
// crossing test, CCW polygon).  For a convex polygon only the two hull edges that straddle the
// horizontal through the centre can toggle, so the answer is "centre.x lies in [xmin, xmax]",
// the intersection of the hull with that line under Agg's half-open rule (y >= ty counts as
// above).  xmin/xmax are extrema of pairwise segment intercepts, which turns into sign tests on
// integer cross products -- no hull, no arrays.  Validated against the reference on all 75 018
// masks of footprints <= 4x4 and 30 000 sampled masks up to 6x6 (tests/golden/stable3d.npz).
__device__ inline int tap_stable3d(int bx, int by, u64 m)
{
    const int k = __popcll(m);
    if (2 * k > bx * by) return 1; // :730
    if (k <= 1) return 0;          // :732
    const int tx = bx - 1, ty = by - 1; // centre, doubled, footprint-local
    const int b0 = __ffsll((long long)m) - 1;
    const int p0x = 2 * (b0 >> 3), p0y = 2 * (b0 & 7);
    u64 rest = m & (m - 1);
    const int b1 = __ffsll((long long)rest) - 1;
    const int p1x = 2 * (b1 >> 3), p1y = 2 * (b1 & 7);
    if (k == 2) return tap_line_rule(tx, ty, p0x, p0y, p1x, p1y); // :734-744
    // all collinear -> qhull raises -> segment rule on (first min-x point, first max-x point)
    bool col = true;
    for (u64 r = rest & (rest - 1); r; r &= r - 1) {
        const int b = __ffsll((long long)r) - 1;
        const int px = 2 * (b >> 3), py = 2 * (b & 7);
        if ((p1x - p0x) * (py - p0y) - (p1y - p0y) * (px - p0x) != 0) col = false;
    }
    if (col) { // :750-762
        const int imax = (63 - __clzll((long long)m)) >> 3;
        const int jlo = __ffsll((long long)((m >> (8 * imax)) & 0xffull)) - 1;
        return tap_line_rule(tx, ty, p0x, p0y, 2 * imax, 2 * jlo);
    }
    bool le = false, ge = false; // exists pair with intercept <= tx / >= tx   (:764-765)
    for (u64 mu = m; mu; mu &= mu - 1) {
        const int bu = __ffsll((long long)mu) - 1;
        const int ux = 2 * (bu >> 3), uy = 2 * (bu & 7);
        if (uy < ty) continue; // upper set: y >= ty
        for (u64 ml = m; ml; ml &= ml - 1) {
            const int bl = __ffsll((long long)ml) - 1;
            const int lx = 2 * (bl >> 3), ly = 2 * (bl & 7);
            if (ly >= ty) continue; // lower set: y < ty
            const int g = (ty - uy) * (lx - ux) - (tx - ux) * (ly - uy);
            le |= g >= 0;
            ge |= g <= 0;
        }
    }
    return le && ge;
}

// Table form of tap_stable3d for footprints up to 4x4 (every RAND block: sides 1..4): one bit per
// (shape, support mask), built once per context by running tap_stable3d itself over every mask of every shape
// (env.hip).  In the 3D kernels the hull test was more than half of a placement's time (7.5 of 14.6 us at
// config c5); the cheap cases (majority, <= 1 support cell) stay inline, the rest become one 32-bit load.
// Layout: shape (bx, by) owns the 64 Kbit slot ((bx-1)*4 + by-1) << 16, and a mask is indexed by its four rows'
// low nibbles packed side by side (the footprint's rows sit 8 bits apart in the stride-8 mask) -- a fixed
// layout, so the index is a handful of 32-bit operations with no per-shape offset table or row loop
// (128 KB, of which only the lines of masks that occur are ever touched).
constexpr int TAP_LUT_SHAPES = 16;
constexpr int TAP_LUT_BITS = TAP_LUT_SHAPES << 16;
constexpr int TAP_LUT_WORDS = TAP_LUT_BITS / 32;

__device__ __forceinline__ unsigned tap_lut_pack(u64 m)   // stride-8 mask of a footprint <= 4x4 -> 16-bit index
{
    unsigned t = (unsigned)m & 0x0f0f0f0fu;                 // rows 0..3 live in the low word
    t = (t | (t >> 4)) & 0x00ff00ffu;
    return (t | (t >> 8)) & 0xffffu;
}
__device__ __forceinline__ u64 tap_lut_unpack(unsigned idx) // inverse of tap_lut_pack
{
    return (u64)((idx & 0xfu) | ((idx & 0xf0u) << 4) | ((idx & 0xf00u) << 8) | ((idx & 0xf000u) << 12));
}

__device__ __forceinline__ int tap_stable3d_any(const uint32_t *lut, int bx, int by, u64 m)
{
    if (lut == nullptr || bx > 4 || by > 4) return tap_stable3d(bx, by, m);
    const int k = __popc((unsigned)m);                       // footprint <= 4x4: the mask is in the low word
    if (2 * k > bx * by) return 1; // tools.py:730
    if (k <= 1) return 0;          // tools.py:732
    const unsigned idx = (unsigned)(((bx - 1) * 4 + (by - 1)) << 16) | tap_lut_pack(m);
    return (lut[idx >> 5] >> (idx & 31u)) & 1u;
}

// ---- footprint scan over the group's LDS slice ---------------------------------------------
// -> mx = max height, eq = cells at that height (2D: bit i; 3D: bit i*8+j), sum = sum of heights
template <int D>
__device__ __forceinline__ void tap_scan(const int *s, int L, int x, int y, int bx, int by,
                                         int &mx, u64 &eq, int &sum)
{
    mx = -1; eq = 0; sum = 0;
    if (D == 2 && bx <= 4) {
        // every RAND block: the (at most 4) LDS reads are issued together, as in the 3D form below
        int h[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = i < bx ? s[x + i] : -1;
#pragma unroll
        for (int i = 0; i < 4; ++i) { mx = max(mx, h[i]); sum += max(h[i], 0); }
        unsigned e = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) e |= (unsigned)(h[i] == mx) << i;   // h = -1 outside the footprint, mx >= 0
        eq = e;
    } else if (D == 2) {
        for (int i = 0; i < bx; ++i) {
            const int h = s[x + i];
            sum += h;
            if (h > mx) { mx = h; eq = 1ull << i; }
            else if (h == mx) eq |= 1ull << i;
        }
    } else if (bx <= 4 && by <= 4) {
        // every RAND block: all (at most 16) LDS reads are issued before the first is used -- the
        // general loop below waits for each read in turn, which was half of a 3D placement's time
        int h[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) h[i][j] = (i < bx && j < by) ? s[(x + i) * L + y + j] : -1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) { mx = max(mx, h[i][j]); sum += max(h[i][j], 0); }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) eq |= (u64)(h[i][j] == mx && i < bx && j < by) << (i * 8 + j);
    } else {
        for (int i = 0; i < bx; ++i)
            for (int j = 0; j < by; ++j) {
                const int h = s[(x + i) * L + y + j];
                sum += h;
                const u64 bit = 1ull << (i * 8 + j);
                if (h > mx) { mx = h; eq = bit; }
                else if (h == mx) eq |= bit;
            }
    }
}

// C+P+S of one settled candidate, fp64 in the reference's operation order
// (tools.py:2124-2140 / 2300-2315; ratio = (c + p) + s at :2161).
__device__ __forceinline__ double tap_score(const PlaceCfg &c, const Counters &cnt, int vol,
                                            int gmax, int z, int bz, int emp, int stab)
{
    const int valid2 = cnt.valid + vol;
    const int height = max(gmax, z + bz);
    const double C = (double)valid2 / (double)((long long)height * c.W * c.L);
    const double P = (c.flags & TAP_F_USE_P) ? (double)valid2 / (double)(emp + valid2) : 0.0;
    const double S = (c.flags & TAP_F_USE_S)
                         ? (double)(cnt.nstable + stab) / (double)(cnt.count + 1) : 0.0;
    return (C + P) + S;
}

// cell / L for the lane-per-cell groups (cell < 64; 3D sides <= 8): a multiplication by ceil(2^16 / L), the
// constant picked by scalar compares -- an integer division by a run-time value is ~25 vector instructions
__device__ __forceinline__ int tap_div_small(int cell, int L)
{
    int m;
    switch (L) {
    case 1: return cell;
    case 2: return cell >> 1;
    case 3: m = 21846; break;
    case 4: return cell >> 2;
    case 5: m = 13108; break;
    case 6: m = 10923; break;
    case 7: m = 9363; break;
    case 8: return cell >> 3;
    default: return cell / L;
    }
    return (cell * m) >> 16;
}

// ---- one placement -------------------------------------------------------------------------
// s        : the group's LDS slice holding the current height-map (written + barrier'd by caller)
// cell     : lane index inside the group; hm : this lane's cell height (updated on commit)
// do_step  : group-uniform; false leaves the env untouched
// err      : per-lane error bits, OR-reduced by the caller (1 = height overflow)
// HARDOK = false compiles the soft form only (the caller has checked that TAP_F_HARD is not set): the hard-mode
// walk is what drives the register count (3D: 106 VGPRs with it, 58 without).
template <int D, int G, bool HARDOK = true>
__device__ inline Placement tap_place(const PlaceCfg &c, const int *s, int cell, int &hm,
                                      Counters &cnt, int &err, int bx, int by, int bz,
                                      bool do_step)
{
    const int W = c.W, L = c.L;
    const int x = (D == 2) ? cell : tap_div_small(cell, L);
    const int y = (D == 2) ? 0 : cell - x * L;
    const bool incell = cell < W * L;
    const bool hard = (c.flags & TAP_F_HARD) != 0;
    const int vol = bx * by * bz;
    const int gmax = group_max<G>(incell ? hm : 0);

    // left-bottom corners from the height-map (tools.py:2067-2078 2D; 2219-2246 3D, appendix B)
    bool corner = false;
    int cls = 0;
    if (incell) {
        if (D == 2) {
            corner = (x == 0) || (hm != s[cell - 1]);
        } else {
            const int h = hm;
            const int hxm = x > 0 ? s[cell - L] : 0;                // hm[x-1, y]
            const int hym = y > 0 ? s[cell - 1] : 0;                // hm[x, y-1]
            const int hxym = (x > 0 && y > 0) ? s[cell - L - 1] : 0; // hm[x-1, y-1]
            const int hxmm = x > 1 ? s[cell - 2 * L] : 0;           // hm[x-2, y]
            const int dx = x > 0 ? h - hxm : 0;                     // hm_diff_x[x, y]
            const int dx_ym = (x > 0 && y > 0) ? hym - hxym : 0;    // hm_diff_x[x, y-1]
            const int dx_xm = x > 1 ? hxm - hxmm : 0;               // hm_diff_x[x-1, y]
            const int dy = y > 0 ? h - hym : 0;                     // hm_diff_y[x, y]
            const int dy_xm = (x > 0 && y > 0) ? hxm - hxym : 0;    // hm_diff_y[x-1, y]
            const bool rej1 = (y > 0) && (dx_ym != 0) && (h == hym) && (dx == dx_ym); // :2234-2237
            const bool c1 = (dx != 0) && !rej1;
            const bool rej2 = (x > 0) && (dy_xm != 0) && (h == hxm) && (dx == dx_xm); // :2241-2244 (sic dx)
            const bool c2 = !c1 && (dy != 0) && !rej2;
            if (cell == 0) { corner = true; cls = 0; }
            else if (c1) { corner = true; cls = 1; }
            else if (c2) { corner = true; cls = 2; }
        }
    }
    const bool cand = do_step && corner && (x + bx <= W) && (y + by <= L); // :2076, :2255-2256

    Placement res = {0, 0, 0, 0, 0};
    int emp_w = 0;

    if (!HARDOK || !hard) {
        // soft: every in-bounds corner settles on itself (z = max of its footprint)
        double ratio = -1.0;
        int key = INT_MAX, z = 0, stab = 0, emp = 0;
        if (cand) {
            int mx, sum; u64 eq;
            tap_scan<D>(s, L, x, y, bx, by, mx, eq, sum);
            z = mx;
            if (z >= c.H) err |= 1;                               // :2109 would raise IndexError
            stab = (z == 0) ? 1 : (D == 2 ? tap_stable2d(bx, eq) : tap_stable3d_any(c.lut, bx, by, eq));
            emp = cnt.empty + bx * by * z - sum;                  // :2132-2134
            ratio = tap_score(c, cnt, vol, gmax, z, bz, emp, stab);
            key = ((z * L + y) * 3 + cls) * W + x;                // sort order (z, y, class, x)
        }
        // argmax over the group on (ratio desc, key asc) -- keys of candidates are unique -- then the winner's
        // placement is fetched from its lane (three cross-lane reads instead of three more values in every step)
        const int lane = (int)(threadIdx.x & 63);
        const int mykey = key;
        group_butterfly<G>(lane, [&](auto get) {
            const double r2 = __hiloint2double(get(__double2hiint(ratio)), get(__double2loint(ratio)));
            const int k2 = get(key);
            if (r2 > ratio || (r2 == ratio && k2 < key)) { ratio = r2; key = k2; }
        });
        res.placed = ratio > 0.0;
        const u64 wm = __ballot(cand && mykey == key);            // each group's winner (none: no candidate)
        const u64 mine = G == 64 ? wm : ((wm >> (lane & ~(G - 1))) & ((1ull << (G & 63)) - 1ull));
        const int src = (lane & ~(G - 1)) + (mine ? __ffsll((long long)mine) - 1 : 0);
        const int pxy = __shfl(x | (y << 8) | (stab << 16), src);
        res.x = pxy & 255;
        res.y = (pxy >> 8) & 255;
        res.z = __shfl(z, src);
        res.stab = pxy >> 16;
        emp_w = __shfl(emp, src);
    } else {
        // hard: the reference walks the sorted corner list sequentially with a shared `visited` set, sliding each
        // block over the positions (_x >= X0, _y >= Y0) until it is supported, free and stable (tools.py:2100-2121,
        // 2284-2297, 2320-2327).  Whether a block settles at a position depends on the position and the corner's
        // level only, so every lane scans the footprint at ITS cell once; a corner's walk is then three ballots
        // (reachable, settles, and the cells up to the first settle become visited), the corner's lane keeps where
        // its walk ended, and the candidates are scored in parallel afterwards -- "first maximum in list order" is
        // the soft branch's (ratio desc, key asc) argmax.  (Before: all lanes walked redundantly, one footprint scan
        // per visited position and three fp64 divisions per settled corner, 0.6 ms per generator launch.)
        const int lane = (int)(threadIdx.x & 63), gl0 = lane & ~(G - 1);
        auto gballot = [&](bool p) -> u64 {
            const u64 b = __ballot(p);
            return G == 64 ? b : ((b >> gl0) & ((1ull << (G & 63)) - 1ull));
        };
        const bool fits = do_step && incell && (x + bx <= W) && (y + by <= L);
        int mx0 = -1, sum0 = 0;
        u64 eq0 = 0;
        if (fits) tap_scan<D>(s, L, x, y, bx, by, mx0, eq0, sum0);
        const int st0 = (fits && mx0 > 0) ? (D == 2 ? tap_stable2d(bx, eq0) : tap_stable3d_any(c.lut, bx, by, eq0)) : 1;
        const int mykey = cand ? ((mx0 * L + y) * 3 + cls) * W + x : INT_MAX;
        const int mine = x | (y << 8) | (max(mx0, 0) << 16);
        int last = -1, vis_z = -1, my_f = -1;
        u64 visited = 0;
        for (int it = 0; it < G; ++it) { // wave-uniform trip count: shuffles stay convergent
            const int kmin = group_min<G>(mykey > last ? mykey : INT_MAX);
            if (__ballot(kmin != INT_MAX) == 0ull) break;   // every group of the wave has walked its last corner
            if (kmin == INT_MAX) continue;
            last = kmin;
            const u64 cm = gballot(mykey == kmin);           // the corner's lane (keys are unique)
            const int cxyz = __shfl(mine, gl0 + __ffsll((long long)cm) - 1);
            const int X0 = cxyz & 255, Y0 = (cxyz >> 8) & 255, z = cxyz >> 16;
            if (z != vis_z) { vis_z = z; visited = 0; }
            const bool reach = fits && x >= X0 && y >= Y0 && !((visited >> cell) & 1ull) && !(z > 0 && mx0 < z); // :2105-2106
            const bool settle = reach && z < c.H && mx0 <= z && (z == 0 || st0);                                 // :2109-2114
            const u64 okm = gballot(settle), reachm = gballot(reach);
            const int f = okm ? __ffsll((long long)okm) - 1 : 64;                     // scan order = cell order
            visited |= reachm & (f >= 63 ? ~0ull : ((2ull << f) - 1ull));             // :2107, nothing past the settle
            if (z >= c.H && reachm) err |= 1;                                         // :2109 IndexError
            if (mykey == kmin) my_f = okm ? f : -1;
        }
        // the settled corners, scored by their own lanes
        const int fsrc = gl0 + max(my_f, 0);
        const int f_xy = __shfl(mine, fsrc), f_sum = __shfl(sum0, fsrc), f_st = __shfl(st0, fsrc);
        double ratio = -1.0;
        int key = INT_MAX, z = 0, stab = 0, emp = 0;
        if (my_f >= 0) {
            z = max(mx0, 0);
            stab = (z == 0) ? 1 : f_st;
            emp = cnt.empty + bx * by * z - f_sum;
            ratio = tap_score(c, cnt, vol, gmax, z, bz, emp, stab);
            key = mykey;
        }
        const int wkey = key;
        group_butterfly<G>(lane, [&](auto get) {
            const double r2 = __hiloint2double(get(__double2hiint(ratio)), get(__double2loint(ratio)));
            const int k2 = get(key);
            if (r2 > ratio || (r2 == ratio && k2 < key)) { ratio = r2; key = k2; }
        });
        res.placed = ratio > 0.0;
        const u64 wm = gballot(my_f >= 0 && wkey == key);
        const int src = gl0 + (wm ? __ffsll((long long)wm) - 1 : 0);
        const int pxy = __shfl((f_xy & 0xffff) | (stab << 16), src);
        res.x = pxy & 255;
        res.y = (pxy >> 8) & 255;
        res.z = __shfl(z, src);
        res.stab = pxy >> 16;
        emp_w = __shfl(emp, src);
    }

    // commit (tools.py:2167-2174); a failed placement leaves everything but the step counter
    if (do_step) {
        if (res.placed) {
            if (incell && x >= res.x && x < res.x + bx && y >= res.y && y < res.y + by) hm = res.z + bz;
            cnt.valid += vol;
            cnt.empty = emp_w;
            cnt.nstable += res.stab;
            if (res.z + bz > c.H) err |= 1;                       // :2169 numpy clips silently
        } else {
            res.x = res.y = res.z = res.stab = 0;
        }
        cnt.count += 1;                                           // tools.py:3713
    }
    return res;
}

// ---- feature writer (tools.py:3716-3744), shared by step, transition and get_heightmap ----------

// s = group's LDS slice with the CURRENT height-map (barrier'd); writes this lane's share
template <int D, int G>
__device__ __forceinline__ void tap_write_feature(int feature, int W, int L, const int *s, int cell,
                                                  int hm, float *out /* env's row */)
{
    const int cells = W * L;
    const bool incell = cell < cells;
    if (feature == TAP_FEAT_DIFF) {
        if (D == 2) {
            if (cell < W - 1) out[cell] = (float)(s[cell + 1] - hm);               // :3739-3743
        } else if (incell) {
            const int x = tap_div_small(cell, L), y = cell - x * L;
            out[cell] = (float)(x > 0 ? hm - s[cell - L] : 0);                      // :3723-3725
            out[cells + cell] = (float)(y > 0 ? hm - s[cell - 1] : 0);              // :3728-3730
        }
    } else if (feature == TAP_FEAT_ZERO) {
        const int mn = group_min<G>(incell ? hm : INT_MAX);                         // :3719
        if (incell) out[cell] = (float)(hm - mn);
    } else if (incell) {
        out[cell] = (float)hm;                                                      // :3717
    }
}


// Container.calc_ratio's formula table (tools.py:3919-3964)
__device__ __forceinline__ double tap_ratio_formula(int mode, double C, double P, double S)
{
    switch (mode) {
    case TAP_R_C: return C / 3;
    case TAP_R_CxS: return (C * S) / 3;
    case TAP_R_CP: return (C + P) / 3;
    case TAP_R_CPxS: return ((C + P) * S) / 3;
    case TAP_R_2CPS: return ((2 * C + P) + S) / 3;
    case TAP_R_CxPxS: return ((C * P) * S) / 3;
    case TAP_R_CP_HALF: return (C + P) / 2;
    default: return ((C + P) + S) / 3;
    }
}

// LDS hand-off between lanes of ONE wavefront (a lane group never spans waves): the LDS unit
// executes a wave's DS instructions in order, so only the compiler must be kept from reordering.
__device__ __forceinline__ void tap_wave_lds_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
