// tap_stable_wide.h -- tools.is_stable (tools.py:710-765) for block footprints beyond the 8 x 8 support masks of
// tap_place.h (tap_stable3d), up to 16 x 16: the big-container kernels (big.hip, tap_macs3_wave.h, tap_macs3_big.h) call
// it when a block side exceeds 8 -- e.g. 10-wide blocks in a 20 x 20 container, which the reference accepts
// (--max_size / --unit are free, generate.py:795).  Same integer form as tap_stable3d: majority,
// <= 1 point, the two-point line rule, the collinear fall-back, and for a proper hull the Agg crossing test reduced to
// "the centre's x lies between the hull's two intercepts with the horizontal through the centre".  That intercept test
// is an existence statement over (upper point, lower point) pairs whose left side is monotone in both x coordinates,
// so only each row's smallest and largest supported x matter: by rows, packed four bits each into two 64-bit words.
// Host-compilable (tests/host: checked against the oracle's hull-based restatement on random masks).
#pragma once

#if defined(__HIPCC__) || defined(__CUDACC__)
#define TAP_SW_HD __host__ __device__
#else
#define TAP_SW_HD
#endif

constexpr int TAP_WIDE_MAX_SIDE = 16;

TAP_SW_HD inline int tap_sw_line_rule(int cx, int cy, int p0x, int p0y, int p1x, int p1y)   // tools.py:736-744 / 755-762, doubled
{
    const int a = cx - p0x, b = cy - p0y, c = cx - p1x, d = cy - p1y;
    if (b == 0 || d == 0) return (b == d) && ((a < 0) != (c < 0));
    return (a * d == c * b) && ((a < 0) != (c < 0)) && ((b < 0) != (d < 0));
}

// height(i, j) = height-map value under footprint cell (i, j); z = the level the block would rest on (> 0)
template <class HeightAt>
TAP_SW_HD inline int tap_stable3d_wide(HeightAt height, int bx, int by, int z)
{
    typedef unsigned long long u64;
    int k = 0, p0x = 0, p0y = 0, p1x = 0, p1y = 0, imax = -1, jlo = 0;
    bool col = true;
    u64 rmin = 0, rmax = 0;                                  // per row j (4 bits each): smallest / largest supported i
    unsigned has = 0;
    for (int i = 0; i < bx; ++i)                             // points in the reference's order: i outer, j inner (:726-729)
        for (int j = 0; j < by; ++j) {
            if (height(i, j) != z) continue;
            const int px = 2 * i, py = 2 * j;
            if (k == 0) { p0x = px; p0y = py; }
            else if (k == 1) { p1x = px; p1y = py; }
            else if ((p1x - p0x) * (py - p0y) - (p1y - p0y) * (px - p0x) != 0) col = false;
            if (i > imax) { imax = i; jlo = j; }             // argmax of x, first occurrence (:753)
            if (!((has >> j) & 1u)) { has |= 1u << j; rmin |= (u64)i << (4 * j); }
            rmax = (rmax & ~(15ull << (4 * j))) | ((u64)i << (4 * j));
            ++k;
        }
    if (2 * k > bx * by) return 1;                           // :730
    if (k <= 1) return 0;                                    // :732
    const int tx = bx - 1, ty = by - 1;                      // centre, doubled, footprint-local
    if (k == 2) return tap_sw_line_rule(tx, ty, p0x, p0y, p1x, p1y);              // :734-744
    if (col) return tap_sw_line_rule(tx, ty, p0x, p0y, 2 * imax, 2 * jlo);       // :750-762 (qhull raises on collinear sets)
    bool le = false, ge = false;                             // :764-765: an intercept <= tx / >= tx exists
    for (int ju = 0; ju < by; ++ju) {
        const int uy = 2 * ju;
        if (!((has >> ju) & 1u) || uy < ty) continue;        // upper set: y >= ty
        const int umin = 2 * (int)((rmin >> (4 * ju)) & 15ull), umax = 2 * (int)((rmax >> (4 * ju)) & 15ull);
        for (int jl = 0; jl < by; ++jl) {
            const int ly = 2 * jl;
            if (!((has >> jl) & 1u) || ly >= ty) continue;   // lower set: y < ty
            const int lmin = 2 * (int)((rmin >> (4 * jl)) & 15ull), lmax = 2 * (int)((rmax >> (4 * jl)) & 15ull);
            // g(u, l) = (ty - uy)(lx - ux) - (tx - ux)(ly - uy) falls in both lx and ux: its largest value over the two
            // rows is at their smallest x, its smallest at their largest
            le |= (ty - uy) * (lmin - umin) - (tx - umin) * (ly - uy) >= 0;
            ge |= (ty - uy) * (lmax - umax) - (tx - umax) * (ly - uy) <= 0;
        }
    }
    return le && ge;
}
