// tap_macs3.h -- device code: one MACS / MUL 3D placement for one container, G lanes per container
// (lane = height-map cell).  tools.calc_one_position_mcs_3d (tools.py:2751-3165) re-stated on the
// height-map, the placement history and ceil(H/64) 64-bit words per cell; used by macs.hip and transition.hip.
//
// State.  The reference keeps a voxel grid and, per (level z, row y), a list of free x-intervals.
//   * voxel (x,y,z) == 0  <=>  z >= hm[x,y]: update_container (tools.py:3043-3047) fills the block
//     and marks everything below it, so the height-map carries every voxel test of the function.
//   * the interval lists are NOT a function of the height-map: update_level_free_space
//     (tools.py:2989-3041) forgets to shrink an interval of a lower level when the new block's
//     shadow falls strictly inside it, so a list can keep cells that are no longer free.  The lists
//     are, however, always the maximal runs of a bit-grid F(x,y,z) with the update rule
//         levels [z, z+bz):  clear the footprint;
//         levels below z, per footprint row: clear the row's cells unless F also holds both
//         x-neighbours of the row AND every cell of the row (the "strictly inside" case)
//     (checked against the reference's lists over 2.3e5 rows, scratch notes in DESIGN.md).  Each
//     cell carries its column of F as ceil(H/64) u64 (bit z), stored complemented so that a zeroed state
//     blob is the empty container; ceil(H/64) <= 8 words per cell, hence H <= 512.
//   * voxel *values* (which block) are only compared in the "partly covered top" case
//     (tools.py:2924-2942) and are recomputed from the placement history when that case occurs.
//
// Structure, as in tap_macs.h: (1) the EMS list is built by every lane of the group redundantly
// (identical LDS writes), level masks come from one ballot each, rows are W-bit fields of a
// y-major mask; the reference's function-scope variable `x1` that the "left part" case reads stale
// (tools.py:2865) is carried exactly; (2) whether a block settles at (x,y,Z) depends on the position
// only, and a position can only settle at Z = max of the height-map under it, so the four corner
// walks of every EMS are find-first-set operations on good(Z) & ~taken & rectangle in the walk's
// order, and `visited` reduces to one `taken` bit per position; (3) each settled position is scored
// by its own lane in fp64; the usable-space tie-break (tools.py:3049-3077) is the sum over levels
// of the largest free rectangle, computed per tied candidate from one ballot per distinct height.
#pragma once

#include "tap_common.h"
#include "tap_place.h"

constexpr int MACS3_EMS_CAP = 192; // packed EMS entries per env (<= 61 seen at 8x8, 40 blocks)
constexpr int MACS3_MAX_H = 4096;  // HW = ceil(H / 64) words per cell, in LDS and in the state blob; no register array scales with it
constexpr int MACS3_HIST = 2;      // ints per history entry: x | y<<4 | xx<<8 | yy<<12 | placed<<16,  z | zz<<16

__host__ __device__ constexpr int macs3_hw(int H) { return (H + 63) / 64; }

// LDS words per env group: occ u64[G*HW] | lvm u64[G+2] | hm[G] | ord[G] | lvh[G+2] | lvr[G+2] | ems[CAP] |
// lrun u8[256] | cand[12 G] | hist[2 n_max]   (lv*: the distinct levels of the height-map, at most cells + 1 of them;
// lrun: longest run of ones of every byte)
__host__ __device__ constexpr int macs3_group_words(int G, int n_max, int H)
{
    return 2 * G * macs3_hw(H) + 2 * (G + 2) + G + G + 2 * (G + 2) + MACS3_EMS_CAP + 64 + 12 * G + MACS3_HIST * n_max;
}

struct Macs3Lds {
    u64 *occ, *lvm;
    int *hm, *ord, *lvh, *lvr, *ems, *cand, *hist;
    unsigned char *lrun;
};

__device__ __forceinline__ Macs3Lds macs3_lds(int *base, int G, int H)
{
    Macs3Lds m;
    const int HW = macs3_hw(H);
    m.occ = reinterpret_cast<u64 *>(base);           // cell-major: word w of cell c at occ[c*HW + w]
    m.lvm = m.occ + G * HW;
    m.hm = base + 2 * G * HW + 2 * (G + 2);
    m.ord = m.hm + G;
    m.lvh = m.ord + G;
    m.lvr = m.lvh + G + 2;
    m.ems = m.lvr + G + 2;
    m.lrun = reinterpret_cast<unsigned char *>(m.ems + MACS3_EMS_CAP);
    m.cand = m.ems + MACS3_EMS_CAP + 64;     // EMS candidates of one round, in list order (phase 1)
    m.hist = m.cand + 12 * G;
    return m;
}

template <int G> __device__ __forceinline__ u64 ballot_g(bool p, int gl0)
{
    const u64 b = __ballot(p);
    if (G == 64) return b;
    if (G == 32) return gl0 ? (b >> 32) : (b & 0xffffffffull);   // a select of two scalar halves, not a 64-bit vector shift
    return (b >> gl0) & ((1ull << (G & 63)) - 1ull);
}

template <int G> __device__ __forceinline__ u64 group_or64(u64 v)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    group_butterfly<G>((int)(threadIdx.x & 63), [&](auto get) { lo |= (unsigned)get((int)lo); hi |= (unsigned)get((int)hi); });
    return ((u64)hi << 32) | lo;
}

// cell / position masks: 32-bit arithmetic when the container has at most 32 cells (variable 64-bit
// shifts run at quarter rate), 64-bit otherwise
template <int G> struct m3_mask { typedef u64 type; };
template <> struct m3_mask<8> { typedef unsigned type; };
template <> struct m3_mask<16> { typedef unsigned type; };
template <> struct m3_mask<32> { typedef unsigned type; };
__device__ __forceinline__ int m3_ffs(unsigned v) { return __ffs((int)v) - 1; }               // -1 if empty
__device__ __forceinline__ int m3_ffs(u64 v) { return __ffsll((long long)v) - 1; }
__device__ __forceinline__ int m3_fls(unsigned v) { return 31 - __clz((int)v); }
__device__ __forceinline__ int m3_fls(u64 v) { return 63 - __clzll((long long)v); }
__device__ __forceinline__ int m3_popc(unsigned v) { return __popc(v); }
__device__ __forceinline__ int m3_popc(u64 v) { return __popcll(v); }

__device__ __forceinline__ unsigned m3_bits(int a, int b) // bits a..b inclusive, empty when a > b
{
    return a > b ? 0u : (((2u << (b - a)) - 1u) << a);
}
__device__ __forceinline__ bool m3_bit(unsigned r, int i) { return (r >> i) & 1u; }
// list semantics on a row mask: is [x1, x2] (bits `run`) exactly one interval of the row?
__device__ __forceinline__ bool m3_has_run(unsigned r, unsigned run, int x1, int x2)
{
    return (r & run) == run && !(x1 > 0 && m3_bit(r, x1 - 1)) && !m3_bit(r, x2 + 1);
}
// `v in list`: v is the first or the last cell of an interval
__device__ __forceinline__ bool m3_inlist(unsigned r, int v)
{
    return m3_bit(r, v) && (v == 0 || !m3_bit(r, v - 1) || !m3_bit(r, v + 1));
}
// longest run of ones in a byte (L <= 8 columns)
__device__ __forceinline__ int m3_longest_run(unsigned v)
{
    int r = 0;
    while (v) { v &= v << 1; ++r; }
    return r;
}
// largest all-free axis-aligned rectangle of an x-major W x L bit grid (bit x*L + y)
// lrun: the group's 256-entry table of m3_longest_run in LDS (one read instead of a data-dependent loop)
// Every (first row, last row) pair, no early exits: the table look-ups are independent of each other and go out
// back to back (a pruned loop waits one LDS round trip per pair).  WW = the container's W as a compile-time bound
// (5 x 5: 15 pairs instead of the 36 of an 8-row grid), picked by a scalar switch.
template <int WW>
__device__ __forceinline__ int m3_maxrect_t(u64 fm, int L, unsigned lmask, const unsigned char *lrun)
{
    unsigned row[WW];
#pragma unroll
    for (int i = 0; i < WW; ++i) row[i] = (unsigned)(fm >> (i * L)) & lmask;
    int best = 0;
#pragma unroll
    for (int i1 = 0; i1 < WW; ++i1) {
        unsigned acc = lmask;
#pragma unroll
        for (int i2 = i1; i2 < WW; ++i2) {
            acc &= row[i2];
            best = max(best, (i2 - i1 + 1) * (int)lrun[acc]);
        }
    }
    return best;
}
__device__ inline int m3_maxrect(u64 fm, int W, int L, unsigned lmask, const unsigned char *lrun)
{
    switch (W) {                                             // kernel-uniform
    case 1: return m3_maxrect_t<1>(fm, L, lmask, lrun);
    case 2: return m3_maxrect_t<2>(fm, L, lmask, lrun);
    case 3: return m3_maxrect_t<3>(fm, L, lmask, lrun);
    case 4: return m3_maxrect_t<4>(fm, L, lmask, lrun);
    case 5: return m3_maxrect_t<5>(fm, L, lmask, lrun);
    case 6: return m3_maxrect_t<6>(fm, L, lmask, lrun);
    case 7: return m3_maxrect_t<7>(fm, L, lmask, lrun);
    default: return m3_maxrect_t<8>(fm, L, lmask, lrun);
    }
}

// -DTAP_PROF: shader-clock deltas per phase of the group in lanes 0..G-1 of each workgroup's first wave
#ifdef TAP_PROF
static __device__ unsigned int tap_prof_m3[8192 * 16];
#define M3_PROF(i) do { const long long t_ = clock64(); if (cell == 0 && (threadIdx.x & 63) == 0 && blockIdx.x < 8192) tap_prof_m3[blockIdx.x * 16 + (i)] = (unsigned)(t_ - tp_); tp_ = t_; } while (0)
#define M3_PROF_BEGIN long long tp_ = clock64()
#else
#define M3_PROF(i) do { } while (0)
#define M3_PROF_BEGIN do { } while (0)
#endif

// One placement.  Preconditions: S.hm[cell] = hm, S.occ[cell] = occ (x-major cells, 0 beyond W*L),
// S.hist[0..8*cnt.count) filled, visible to the group (wave-level sync by the caller).  do_step is
// group-uniform.  On return hm/cnt and the cell's words in S.occ are updated and res describes the
// placement.
template <int G, int WL = 0>
__device__ inline Placement tap_macs3_place(const PlaceCfg &c, const Macs3Lds &S, int cell, int gl0, int &hm,
                                            Counters &cnt, int &err, int bx, int by, int bz, bool do_step)
{
    // WL != 0: a WL x WL container with the sides known at compile time (the row / level loops unroll, `y * W` shifts
    // and the divisions by W fold): 33.4 -> 31.1 us per fused step at c6 (5 x 5, the reference's own 3D container)
    const int W = WL ? WL : c.W, L = WL ? WL : c.L, H = c.H, cells = W * L;
    Placement res = {0, 0, 0, 0, 0};
    if (!do_step) return res;
    M3_PROF_BEGIN;
    const int hard = c.flags & TAP_F_HARD;
    const int vol = bx * by * bz, step = cnt.count;
    typedef typename m3_mask<G>::type mk;                // masks over cells / positions
    const int HW = macs3_hw(H);
    const unsigned wmask = (1u << W) - 1u, lmask = (1u << L) - 1u;
    const int wl = threadIdx.x & 63;

    // this lane in the y-major view: position / cell (tx, ty), bit ty*W + tx of every level mask
    const bool inT = cell < cells;
    const int ty = tap_div_small(cell, W), tx = cell - ty * W;
    const int hmT = inT ? S.hm[tx * L + ty] : INT_MAX;
    const u64 *occT = S.occ + (size_t)(inT ? tx * L + ty : 0) * HW;   // this lane's column of F, complemented
    auto wordF = [&](int w) -> u64 {                                     // F bits of levels [64w, 64w + 64)
        const int nb = H - 64 * w;
        return inT ? (~occT[w] & (nb >= 64 ? ~0ull : ((1ull << nb) - 1ull))) : 0ull;
    };
    const int gmax = group_max<G>(inT ? hmT : 0);
    mk colsel = 0; // bit y*W for every row
    for (int y = 0; y < L; ++y) colsel |= (mk)1 << (y * W);

    auto rowT = [&](mk m, int y) -> unsigned { return (unsigned)(m >> (y * W)) & wmask; };
    auto levelT = [&](int z) -> mk { return (mk)ballot_g<G>(hmT <= z, gl0); }; // container[.., z] == 0

    // ---- phase 1: EMS list (identical on every lane of the group) -------------------------------
    int n_ems = 0;
#define M3_PACK(x1, y1, z, x2, y2) (((x1) & 15) | (((y1) & 15) << 4) | (((x2) & 15) << 8) | (((y2) & 15) << 12) | ((z) << 16))
#define M3_PUSH(x1, y1, z, x2, y2)                                                           \
    do {                                                                                     \
        if (n_ems < MACS3_EMS_CAP) S.ems[n_ems++] = M3_PACK(x1, y1, z, x2, y2);              \
        else err |= 16;                                                                      \
    } while (0)
    // y2 := last row reached going +y from `from` while rows keep `span` free at level mask T
#define M3_EXT_UP(T, span, from, y2)                                                         \
    for (y2 = (from);; ++y2) {                                                               \
        if (y2 == L - 1) break;                                                              \
        if ((rowT(T, y2 + 1) & (span)) != (span)) break;                                     \
    }
#define M3_EXT_DOWN(T, span, from, y1)                                                       \
    for (y1 = (from);; --y1) {                                                               \
        if (y1 == 0) break;                                                                  \
        if ((rowT(T, y1 - 1) & (span)) != (span)) break;                                     \
    }
    int sx1 = 0;          // python's function-scope `x1` (tools.py:2823, 2870, 2901 assign it)
    bool x1def = false;   // ... which :2865 may read before any assignment (UnboundLocalError)
    const mk below_me = ((mk)1 << cell) - 1;
    // exclusive prefix sum of v over the group's lanes (and the group's total)
    // (bit by bit from ballots -- v < 2^NB -- instead of a shuffle scan: log2(G) trips through the LDS crossbar)
    auto excl_scan = [&](int v, int &total, int nbits) -> int {
        int pre = 0, tot = 0;
        for (int b = 0; b < nbits; ++b) {                                            // kernel-uniform trip count
            const mk m = (mk)ballot_g<G>((v >> b) & 1, gl0);
            pre += m3_popc((mk)(m & below_me)) << b;
            tot += m3_popc(m) << b;
        }
        total = tot;
        return pre;
    };
    // the last lane (in list order) that assigned `x1` hands it on
    auto carry_x1 = [&](bool asg, int asgv) {
        const mk am = (mk)ballot_g<G>(asg, gl0);
        const int v_ = __shfl(asgv, gl0 + (am ? m3_fls(am) : 0));
        if (am) { sx1 = v_; x1def = true; }
    };

    // (a) per-(level, row) free intervals (tools.py:2813-2841); level z is skipped when all its
    //     lists equal those of z-1 (:2816), i.e. when no cell's F column changes between the two.
    //     What a (level, row) pair contributes depends on the pair alone, so the pairs are evaluated one per
    //     lane -- G / rows levels at a time -- into S.cand (8 slots per lane: <= 4 runs x <= 2 spaces) and
    //     appended in (level, row, run) order with a prefix count.  The level masks come from ballots in a
    //     short group-uniform loop; each lane keeps those of its own level.
    const int zmax = H - bz;                                                         // :2815
    {
        const int Lp = L - by + 1;                       // rows with y + by <= L (:2818)
        const int LPR = G / Lp;                          // levels per round
        const int ls = cell / Lp, yrow = cell - ls * Lp;
        int slot = 0, myz = -1;
        mk myF = 0, myB = 0, myT = 0;
        auto flush = [&]() {
            int cn = 0, asgv = 0;
            bool asg = false;
            if (myz >= 0) {
                const int y = yrow, z = myz;
                const unsigned row = rowT(myF, y), prow = y > 0 ? rowT(myF, y - 1) : 0u;
                if (!(y > 0 && row == prow)) {                                       // :2819
                    const unsigned brow = z > 0 ? rowT(myB, y) : 0u;
                    unsigned m = row;
                    while (m) {
                        const int x1 = __ffs((int)m) - 1;
                        const int len = __ffs((int)~(m >> x1)) - 1;
                        const unsigned run = ((1u << len) - 1u) << x1;
                        const int x2 = x1 + len - 1;
                        m &= ~run;
                        asg = true; asgv = x1;                                       // :2823
                        if (x1 + bx > W) break;                                      // :2824
                        if (y > 0 && m3_has_run(prow, run, x1, x2)) continue;        // :2825-2827
                        if (z > 0 && m3_has_run(brow, run, x1, x2)) continue;        // :2828-2830
                        bool xspace = true;                                          // :2831-2840
                        int y2;
                        for (y2 = y;; ++y2) {
                            if (y2 == L - 1) break;
                            if ((rowT(myT, y2 + 1) & run) != run) break;
                            if (xspace) {
                                const unsigned f = rowT(myF, y2 + 1);
                                if (!(m3_inlist(f, x1) && m3_inlist(f, x2))) { xspace = false; S.cand[cell * 8 + cn++] = M3_PACK(x1, y, z, x2, y2); }
                            }
                        }
                        S.cand[cell * 8 + cn++] = M3_PACK(x1, y, z, x2, y2);
                    }
                }
            }
            int total;
            const int off = excl_scan(cn, total, 4);                                 // cn <= 8
            for (int j = 0; j < cn; ++j) {
                const int at = n_ems + off + j;
                if (at < MACS3_EMS_CAP) S.ems[at] = S.cand[cell * 8 + j];            // own slots: no hand-off needed
                else err |= 16;
            }
            n_ems = min(MACS3_EMS_CAP, n_ems + total);
            carry_x1(asg, asgv);
            slot = 0; myz = -1;
        };
        for (int w = 0; w < HW && 64 * w <= zmax; ++w) {
            const u64 Fw = wordF(w), carry = w > 0 ? (wordF(w - 1) >> 63) : 0ull;
            const u64 Fw1 = (Fw << 1) | carry;                                       // bit zb: F at level 64w + zb - 1
            u64 chg = group_or64<G>(Fw ^ Fw1) | (w == 0 ? 1ull : 0ull);
            if (zmax - 64 * w < 63) chg &= (2ull << (zmax - 64 * w)) - 1ull;
            while (chg) {
                const int zb = __ffsll((long long)chg) - 1, z = 64 * w + zb;
                chg &= chg - 1ull;
                const mk Fz = (mk)ballot_g<G>((Fw >> zb) & 1ull, gl0);               // from registers: no LDS in this loop
                const mk Fb = z > 0 ? (mk)ballot_g<G>((Fw1 >> zb) & 1ull, gl0) : (mk)0, Tz = levelT(z);
                if (ls == slot) { myF = Fz; myB = Fb; myT = Tz; myz = z; }
                if (++slot == LPR) flush();
            }
        }
        if (slot > 0) flush();
        tap_wave_lds_sync();                                                         // the list, for (b)'s look-ups
    }
    M3_PROF(0);
    // (b) spaces next to and on top of the blocks placed so far (tools.py:2843-2942); a block that
    //     could not be placed sits at (0,0,0) in `positions` and is visited all the same.
    //     One lane per placed block (chunks of CB blocks that fit S.cand): the spaces beside a block depend on the block and
    //     the height-map only -- except the stale `x1` of :2865, which is the last assignment of an EARLIER
    //     block and is handed on through a ballot.  Every lane lays its candidates out in list order
    //     (S.cand, offsets from a prefix count); "top" candidates carry a flag: the reference appends them
    //     only when the list does not hold them yet (:2913, :2941), which the ordered compaction below
    //     reproduces (an equal EARLIER candidate is in the list, or equals an entry that is).  Partly
    //     covered tops (:2915-2942) need the voxel identities and stay a group-wide computation per block.
    constexpr int M3_DD = 1 << 30;                       // "append if absent"
    const int CB = min(G, 12 * G / (4 + cells));         // blocks per chunk: <= 4 + footprint <= 4 + cells candidates each
    for (int base = 0; base < step; base += CB) {
        const int nb = min(CB, step - base);
        const bool valid = cell < nb;
        const int2 hb = reinterpret_cast<const int2 *>(S.hist)[base + (valid ? cell : 0)];
        const int x = hb.x & 15, y = (hb.x >> 4) & 15, xx = (hb.x >> 8) & 15, yy = (hb.x >> 12) & 15;
        const int z = hb.y & 0xffff, zz = hb.y >> 16;
        const int xe = x + xx - 1, t = z + zz;
        mk T = 0, Tt = 0;
        for (int k = 0; k < nb; ++k) {                                               // group-uniform
            const int2 hk = reinterpret_cast<const int2 *>(S.hist)[base + k];
            const int zk = hk.y & 0xffff;
            const mk Ta = levelT(zk), Tb = levelT(zk + (hk.y >> 16));
            if (cell == k) { T = Ta; Tt = Tb; }
        }
        const unsigned spanx = m3_bits(x, xe);
        int a0 = -1, a1 = -1, b0 = -1, b1 = -1, asgv = 0, st_x2 = 0;
        bool asg = false, need_stale = false;
        if (valid) {
            if (y + yy < L) {                                                        // :2847 beyond +y
                const unsigned r = rowT(T, y + yy);
                int y2;
                if ((r & spanx) == spanx) {                                          // :2849
                    if ((x > 0 && m3_bit(r, x - 1)) || (x + xx < W && m3_bit(r, x + xx))) {
                        M3_EXT_UP(T, spanx, y + yy, y2);
                        a0 = M3_PACK(x, y + yy, z, xe, y2);
                    }
                } else {
                    if (m3_bit(r, x) && x > 0 && m3_bit(r, x - 1)) {                 // :2858 left part
                        need_stale = true;
                        st_x2 = x + min(__ffs((int)~(r >> x)) - 1, xx) - 1;          // :2860-2862
                    }
                    if (m3_bit(r, xe) && x + xx < W && m3_bit(r, x + xx)) {          // :2868 right part
                        const int down = __clz((int)~(r << (31 - xe)));              // free cells from xe leftwards
                        const int x1 = xe - min(down, xx) + 1;                       // :2870-2872
                        asg = true; asgv = x1;
                        const unsigned sp = m3_bits(x1, xe);
                        M3_EXT_UP(T, sp, y + yy, y2);
                        a1 = M3_PACK(x1, y + yy, z, xe, y2);
                    }
                }
            }
            if (y > 0) {                                                             // :2878 beyond -y
                const unsigned r = rowT(T, y - 1);
                int y1;
                if ((r & spanx) == spanx) {
                    if ((x > 0 && m3_bit(r, x - 1)) || (x + xx < W && m3_bit(r, x + xx))) {
                        M3_EXT_DOWN(T, spanx, y - 1, y1);
                        b0 = M3_PACK(x, y1, z, xe, y - 1);
                    }
                } else {
                    if (m3_bit(r, x) && x > 0 && m3_bit(r, x - 1)) {                 // :2889
                        const int x2 = x + min(__ffs((int)~(r >> x)) - 1, xx) - 1;
                        const unsigned sp = m3_bits(x, x2);                          // :2896 uses x here
                        M3_EXT_DOWN(T, sp, y - 1, y1);
                        b0 = M3_PACK(x, y1, z, x2, y - 1);
                    }
                    if (m3_bit(r, xe) && x + xx < W && m3_bit(r, x + xx)) {          // :2899
                        const int down = __clz((int)~(r << (31 - xe)));
                        const int x1 = xe - min(down, xx) + 1;
                        asg = true; asgv = x1;
                        const unsigned sp = m3_bits(x1, xe);
                        M3_EXT_DOWN(T, sp, y - 1, y1);
                        b1 = M3_PACK(x1, y1, z, xe, y - 1);
                    }
                }
            }
        }
        {   // :2865 reads the `x1` an earlier block (or phase (a)) left behind
            const mk am = (mk)ballot_g<G>(asg, gl0), earlier = am & below_me;
            const int v_ = __shfl(asgv, gl0 + (earlier ? m3_fls(earlier) : 0));
            const int sxl = earlier ? v_ : sx1;
            if (need_stale) {
                if (!(earlier || x1def)) err |= 8;
                const unsigned sp = m3_bits(sxl, st_x2);                             // (sic: stale x1)
                int y2;
                M3_EXT_UP(T, sp, y + yy, y2);
                a0 = M3_PACK(x, y + yy, z, st_x2, y2);
            }
            const int vl = __shfl(asgv, gl0 + (am ? m3_fls(am) : 0));
            if (am) { sx1 = vl; x1def = true; }
        }
        M3_PROF(8);
        const bool has_top = valid && t < H;                                         // :2909 on top
        bool full = true;
        for (int j = 0; j < yy; ++j) full = full && (rowT(Tt, y + j) & spanx) == spanx;
        const int ns = (a0 >= 0) + (a1 >= 0) + (b0 >= 0) + (b1 >= 0);
        int total;
        const int off = excl_scan(ns + (has_top ? (full ? 1 : xx * yy) : 0), total, 7);   // <= 4 + 64
        {
            int j = off;
            if (a0 >= 0) S.cand[j++] = a0;
            if (a1 >= 0) S.cand[j++] = a1;
            if (b0 >= 0) S.cand[j++] = b0;
            if (b1 >= 0) S.cand[j++] = b1;
            if (has_top && full) S.cand[j] = M3_PACK(x, y, t, xe, y + yy - 1) | M3_DD; // :2911-2913
        }
        M3_PROF(9);
        const mk pm = (mk)ballot_g<G>(has_top && !full, gl0);
        if (pm) {
            // Partly covered tops (:2915-2942).  The reference scans the footprint cell by cell; whether cell
            // (i, j) yields a space, and which, depends on the cell alone: the block's lane leaves one marker per
            // footprint cell in its candidate slots, and the markers are then resolved one per lane, across all
            // the chunk's blocks at once.  What a cell needs of its block: the free mask of the top level (Tt,
            // parked in lvh/lvr) and, for :2928/:2934, which cells hold the same voxel value as their -x
            // neighbour at that level (EQ, parked in lvm) -- a value being a block index, -1 below a block, 0 free.
            if (has_top && !full)
                for (int q = 0; q < xx * yy; ++q) S.cand[off + ns + q] = -2 - (cell * 64 + q);
            reinterpret_cast<u64 *>(S.lvh)[cell] = (u64)Tt;
            // voxel identities from the history: colb = placed blocks over this lane's column (tx, ty),
            // alive = placed blocks that cross the top level of this lane's BLOCK; one pass for both
            u64 colb = 0, alive = 0;
            const bool by_bits = step <= 64, narrow = step <= 32;                   // narrow: 32-bit shifts and shuffles
            if (narrow) {
                unsigned cb = 0u, al = 0u;
#pragma unroll 4
                for (int q = 0; q < step; ++q) {
                    const int2 hk = reinterpret_cast<const int2 *>(S.hist)[q];
                    const bool placed = (hk.x >> 16) & 1;
                    const int kz = hk.y & 0xffff, kx = hk.x & 15, ky = (hk.x >> 4) & 15;
                    const bool over = tx >= kx && tx < kx + ((hk.x >> 8) & 15) && ty >= ky && ty < ky + ((hk.x >> 12) & 15);
                    const bool cross = t >= kz && t < kz + (hk.y >> 16);
                    cb |= (unsigned)(placed && over) << q;
                    al |= (unsigned)(placed && cross) << q;
                }
                colb = cb; alive = al;
            } else if (by_bits)
#pragma unroll 4
                for (int q = 0; q < step; ++q) {                                     // no branches: the reads pipeline
                    const int2 hk = reinterpret_cast<const int2 *>(S.hist)[q];
                    const bool placed = (hk.x >> 16) & 1;
                    const int kz = hk.y & 0xffff, kx = hk.x & 15, ky = (hk.x >> 4) & 15;
                    const bool over = tx >= kx && tx < kx + ((hk.x >> 8) & 15) && ty >= ky && ty < ky + ((hk.x >> 12) & 15);
                    const bool cross = t >= kz && t < kz + (hk.y >> 16);
                    colb |= (u64)(placed && over) << q;
                    alive |= (u64)(placed && cross) << q;
                }
            auto shfl64 = [narrow](u64 v, int src) -> u64 {
                const unsigned lo = (unsigned)__shfl((int)v, src);
                return narrow ? (u64)lo : (((u64)(unsigned)__shfl((int)(v >> 32), src) << 32) | lo);
            };
            const int left = (wl + 63) & 63;                                         // the cell at x-1 (same row)
            const u64 colbL = shfl64(colb, left);
            const int hmL = __shfl(hmT, left);
            for (mk pw = pm; pw; pw &= pw - 1) {                                     // group-uniform
                const int k = m3_ffs(pw);
                const int ut = __shfl(t, gl0 + k);
                bool eq;
                if (by_bits) {
                    const u64 al = shfl64(alive, gl0 + k);
                    const u64 mine = colb & al, theirs = colbL & al;
                    eq = mine == theirs && (mine != 0 || (hmT > ut) == (hmL > ut));
                } else {
                    int id = inT ? (hmT > ut ? -1 : 0) : 0;
                    for (int q = 0; q < step; ++q) {
                        const int2 hk = reinterpret_cast<const int2 *>(S.hist)[q];
                        const int kz = hk.y & 0xffff;
                        if (!((hk.x >> 16) & 1) || ut < kz || ut >= kz + (hk.y >> 16)) continue;
                        const int kx = hk.x & 15, ky = (hk.x >> 4) & 15;
                        if (tx >= kx && tx < kx + ((hk.x >> 8) & 15) && ty >= ky && ty < ky + ((hk.x >> 12) & 15)) id = q + 1;
                    }
                    eq = id == __shfl(id, left);
                }
                S.lvm[k] = ballot_g<G>(inT && tx > 0 && eq, gl0);                   // same value from every lane
            }
            tap_wave_lds_sync();
            for (int k0 = 0; k0 < total; k0 += G) {
                const int idx = k0 + cell;
                const int v = idx < total ? S.cand[idx] : 0;
                if (v > -2) continue;
                const int k = (-2 - v) >> 6, fc = (-2 - v) & 63;
                const int2 hu = reinterpret_cast<const int2 *>(S.hist)[base + k];
                const int ux = hu.x & 15, uy = (hu.x >> 4) & 15, uxx = (hu.x >> 8) & 15, uyy = (hu.x >> 12) & 15;
                const int uz = hu.y & 0xffff;
                const mk EQ = (mk)S.lvm[k], Tu = (mk)reinterpret_cast<const u64 *>(S.lvh)[k];
                auto hist = [&](int i, int j) -> int {                               // :2915-2922
                    const unsigned hv_ = (rowT(Tu, uy + j) >> (ux + i)) & ((1u << (uxx - i)) - 1u);
                    return __ffs((int)~hv_) - 1;
                };
                auto rows_equal = [&](int i, int ja, int jb) -> bool {               // rows x+i, x+i-1 over [ja, jb)
                    bool e_ = true;
                    for (int j = ja; j < jb; ++j) e_ = e_ && ((EQ >> ((uy + j) * W + ux + i)) & 1);
                    return e_;
                };
                int want = -1;
                const int i = fc / uyy, j = fc - i * uyy;
                const int hv = hist(i, j);
                bool ok = hv != 0 && !(j > 0 && hv == hist(i, j - 1)) && !(i > 0 && rows_equal(i, j, uyy)); // :2926-2928
                if (ok) {
                    const int i2 = i + hv - 1;
                    int j2, j1;
                    for (j2 = j;; ++j2) { if (j2 == uyy - 1) break; if (hist(i, j2 + 1) < hv) break; }
                    ok = !(i > 0 && rows_equal(i, j, j2));                           // :2934 (empty range is "equal")
                    for (j1 = j;; --j1) { if (j1 == 0) break; if (hist(i, j1 - 1) < hv) break; }
                    if (ok) want = M3_PACK(ux + i, uy + j1, uz, ux + i2, uy + j2) | M3_DD; // :2940 (sic: level z, not z+zz)
                }
                S.cand[idx] = want;
            }
        }
        tap_wave_lds_sync();
        M3_PROF(10);
        // ordered append; flagged candidates only when absent (:2913, :2941)
        for (int k0 = 0; k0 < total; k0 += G) {
            const int k = k0 + cell;
            const int v = k < total ? S.cand[k] : -1;
            const int val = v & ~M3_DD;
            bool keep = v >= 0;
            if (keep && (v & M3_DD)) {
                bool dup = false;
                for (int i = 0; i < n_ems; ++i) dup |= S.ems[i] == val;
                for (int i = k0; i < k; ++i) dup |= (S.cand[i] & ~M3_DD) == val;
                keep = !dup;
            }
            const mk km = (mk)ballot_g<G>(keep, gl0);
            const int at = n_ems + m3_popc((mk)(km & below_me));
            if (keep) { if (at < MACS3_EMS_CAP) S.ems[at] = val; else err |= 16; }
            n_ems = min(MACS3_EMS_CAP, n_ems + m3_popc(km));
            tap_wave_lds_sync();                                                     // entries written by other lanes
        }
    }

    M3_PROF(1);
    // ---- phase 2: the four corner walks of every EMS (tools.py:3080-3115) -------------------------
    // this lane's position (tx, ty): it can only settle at Z = max height under the footprint
    const bool posv = inT && tx + bx <= W && ty + by <= L;
    int mp = -1, sum_p = 0, stab_p = 0;
    if (posv) {
        u64 eq;
        tap_scan<3>(S.hm, L, tx, ty, bx, by, mp, eq, sum_p);
        stab_p = mp == 0 ? 1 : tap_stable3d_any(c.lut, bx, by, eq);                  // tools.is_stable
    }
    const bool okp = posv && (stab_p || !hard);                                      // :2963-2965
    M3_PROF(11);
    const int X = W - bx + 1, Y = L - by + 1;
    // The walk order is sequential (a position settled for one space is skipped by the later ones), but what a
    // space contributes apart from that is not: one lane per space builds the set of positions that settle at
    // the space's level inside each of its four search rectangles -- in row-major bit order for the walks that
    // go by rows, in column-major order for those that go by columns, so that every walk is ONE find-first /
    // find-last on mask & ~taken -- and the sequential pass only reads those masks back.
    S.lvr[cell] = okp ? mp : -1;                         // position -> settling level (lvr is idle until the tie-break)
    S.ord[cell] = -1;
    tap_wave_lds_sync();
    mk taken = 0, takenx = 0;                            // settled positions: bit y*W + x / bit x*L + y
    int n_slots = 0;
    mk colselx = 0;                                      // bit x*L for every column
    for (int xq = 0; xq < W; ++xq) colselx |= (mk)1 << (xq * L);
    auto rect = [&](int xa, int xb, int ya, int yb) -> mk { // [xa, xb) x [ya, yb), bit y*W + x
        const mk rows = colsel & ((yb >= L ? ~(mk)0 : (((mk)1 << (yb * W)) - 1)) & ~(((mk)1 << (ya * W)) - 1));
        return (mk)m3_bits(xa, xb - 1) * rows;
    };
    auto rectx = [&](int xa, int xb, int ya, int yb) -> mk { // the same, bit x*L + y
        const mk cols = colselx & ((xb >= W ? ~(mk)0 : (((mk)1 << (xb * L)) - 1)) & ~(((mk)1 << (xa * L)) - 1));
        return (mk)m3_bits(ya, yb - 1) * cols;
    };
    const int invW = (256 + W - 1) / W, invL = (256 + L - 1) / L;   // p / W == (p * invW) >> 8 for p < 64, W <= 8
    auto settle = [&](int px, int py) {
        const int p = py * W + px;
        taken |= (mk)1 << p;
        takenx |= (mk)1 << (px * L + py);
        S.ord[p] = n_slots++; // same value from every lane of the group
    };
    mk *wm = reinterpret_cast<mk *>(S.cand);             // 5 masks per space of the round
    for (int e0 = 0; e0 < n_ems; e0 += G) {
        mk U = 0, M1 = 0, M2 = 0, M3 = 0, M4 = 0;
        if (e0 + cell < n_ems) {
            const int pk = S.ems[e0 + cell];
            const int X1 = pk & 15, Y1 = (pk >> 4) & 15, X2 = (pk >> 8) & 15, Y2 = (pk >> 12) & 15, Z = pk >> 16;
            const int xr = X2 - bx + 2, yr = Y2 - by + 2;
            mk gy = 0, gx = 0;                           // positions that settle at Z, both bit orders
            for (int py = 0; py < Y; ++py)
                for (int px = 0; px < X; ++px) {
                    const bool hit = S.lvr[py * W + px] == Z;
                    gy |= (mk)hit << (py * W + px);
                    gx |= (mk)hit << (px * L + py);
                }
            if (X1 < X && Y1 < Y) { M1 = gx & rectx(X1, X, Y1, Y); U |= gy & rect(X1, X, Y1, Y); }   // :3085 x up, then y up
            if (xr > 0 && Y1 < Y) { M2 = gy & rect(0, xr, Y1, Y); U |= M2; }                          // :3093 y up, then x down
            if (xr > 0 && yr > 0) { M3 = gx & rectx(0, xr, 0, yr); U |= gy & rect(0, xr, 0, yr); }    // :3101 x down, then y down
            if (X1 < X && yr > 0) { M4 = gy & rect(X1, X, 0, yr); U |= M4; }                          // :3109 y down, then x up
        }
        M3_PROF(12);
        wm[cell * 5] = U; wm[cell * 5 + 1] = M1; wm[cell * 5 + 2] = M2; wm[cell * 5 + 3] = M3; wm[cell * 5 + 4] = M4;
        tap_wave_lds_sync();
        const int nk = min(G, n_ems - e0);
        mk Un = wm[0], n1 = wm[1], n2 = wm[2], n3 = wm[3], n4 = wm[4];
        for (int k = 0; k < nk; ++k) {
            const mk Uk = Un, c1 = n1, c2 = n2, c3 = n3, c4 = n4;
            const int kn = (k + 1 < nk ? k + 1 : k) * 5;                             // next space's masks, in flight
            Un = wm[kn]; n1 = wm[kn + 1]; n2 = wm[kn + 2]; n3 = wm[kn + 3]; n4 = wm[kn + 4];
            if (!(Uk & ~taken)) continue;
            const mk m1 = c1 & ~takenx;
            if (m1) {
                const int q = m3_ffs(m1), px = (q * invL) >> 8;
                settle(px, q - px * L);
            }
            const mk m2 = c2 & ~taken;
            if (m2) {
                const int py = (m3_ffs(m2) * invW) >> 8;
                settle(31 - __clz((int)rowT(m2, py)), py);
            }
            const mk m3 = c3 & ~takenx;
            if (m3) {
                const int q = m3_fls(m3), px = (q * invL) >> 8;
                settle(px, q - px * L);
            }
            const mk m4 = c4 & ~taken;
            if (m4) {
                const int py = (m3_fls(m4) * invW) >> 8;
                settle(__ffs((int)rowT(m4, py)) - 1, py);
            }
        }
        tap_wave_lds_sync();                                                         // the masks are replaced next round
    }
    tap_wave_lds_sync();
    const int ord = S.ord[cell];
    const bool settled = ord >= 0;

    M3_PROF(2);
    // ---- phase 3: score (tools.py:2973-2987), every settled position by its own lane --------------
    const int valid2 = cnt.valid + vol;
    const bool tiebreak = (c.flags & TAP_F_MCS_TIE) != 0, zero = (c.flags & TAP_F_MCS_ZERO) != 0;
    const int emp_p = cnt.empty + bx * by * mp - sum_p;                              // :2982-2983
    double r = -1.0;
    if (settled) {
        if (zero) r = 0.0;                                                           // :3125
        else {
            int height = max(gmax, mp + bz);
            if (mp + bx > height) height = mp + bz;                                  // :2977 (sic block_x)
            const double C = (double)valid2 / (double)((long long)height * W * L);
            const double P = (c.flags & TAP_F_USE_P) ? (double)valid2 / (double)(emp_p + valid2) : 0.0;
            const double S_ = (c.flags & TAP_F_USE_S) ? (double)(cnt.nstable + stab_p) / (double)(cnt.count + 1) : 0.0;
            r = (C + P) + S_;
        }
    }
    const double rmax = group_fmax<G>(r);
    int win = -1; // y-major position of the winner
    M3_PROF(5);
    if (n_slots > 0) {
        const mk tied = (mk)ballot_g<G>(settled && r == rmax, gl0);
        const int nt = zero ? 4 * n_ems : m3_popc(tied); // len(best_ems_indexes), unsettled entries are 0.0
        if (!(tiebreak && nt > 1)) {                                                 // :3145-3148
            const int wo = group_min<G>((settled && r == rmax) ? ord : INT_MAX);
            win = __ffsll((long long)ballot_g<G>(settled && ord == wo, gl0)) - 1;
        } else {                                                                     // :3132-3144
            // calc_maximal_usable_spaces (:3049-3077) = sum over levels h < max_height of the largest
            // free rectangle; max_height is common, above max(hm') a level is all free, so candidates
            // are ordered by  sum_{h < max(hm')} rect(h) - max(hm') * W * L.
            // The current map's distinct levels (height, free mask, largest rectangle) are tabulated
            // once by the group; a candidate's map differs only under its footprint, and only below
            // its top, so every tied lane then sums its own candidate serially from that table.
            const int max_height = max(gmax, group_max<G>(settled ? mp + bz : 0));   // :3133
            if (max_height > H) err |= 1;                                            // container[:, :, h] IndexError
            // levels = 0 and the distinct heights above 0, ascending.  Each cell compares its height with every
            // other cell's (independent LDS reads): the cells at or below it ARE its level's free mask, and its
            // slot in the table is the number of distinct lower levels -- no level-by-level minimum search.
            const bool own = cell < cells;
            const int hx = own ? hm : INT_MAX;                                       // x-major: own cell
            mk le = 0, eq = 0;
            for (int j = 0; j < cells; ++j) {
                const int hj = S.hm[j];
                le |= (mk)(hj <= hx) << j;
                eq |= (mk)(hj == hx) << j;
            }
            const mk ground = (mk)ballot_g<G>(hx <= 0, gl0);                         // level 0's mask
            const bool rep = own && hx > 0 && m3_ffs(eq) == cell;                    // first cell of its height
            const mk reps = (mk)ballot_g<G>(rep, gl0);
            const int nl = 1 + m3_popc(reps);
            S.lvh[0] = 0;                                                            // same values from every lane
            S.lvm[0] = (u64)ground;
            if (rep) {
                const int k = 1 + m3_popc((mk)(le & ~eq & ~ground & reps));              // distinct levels in (0, hx)
                S.lvh[k] = hx;
                S.lvm[k] = (u64)le;
            }
            tap_wave_lds_sync();
            for (int k = cell; k < nl; k += G) S.lvr[k] = m3_maxrect(S.lvm[k], W, L, lmask, S.lrun);
            tap_wave_lds_sync();
            M3_PROF(6);
            int adj = INT_MIN;
            if (settled && r == rmax) {
                const int Zt = mp + bz, M = max(gmax, Zt);
                u64 footm = 0;
                for (int i = 0; i < bx; ++i) footm |= (u64)(((1u << by) - 1u) << ty) << ((tx + i) * L);
                int base = 0;
                for (int k = 0; k < nl; ++k) {
                    const int lo = S.lvh[k];
                    if (lo >= M) break;
                    const int hi = min(M, k + 1 < nl ? S.lvh[k + 1] : INT_MAX);
                    const u64 fm = S.lvm[k];
                    const int a_hi = min(hi, Zt), b_lo = max(lo, Zt);
                    if (a_hi > lo)                                                   // below the block's top
                        base += (a_hi - lo) * ((fm & footm) ? m3_maxrect(fm & ~footm, W, L, lmask, S.lrun) : S.lvr[k]);
                    if (hi > b_lo) base += (hi - b_lo) * S.lvr[k];
                }
                adj = base - M * cells;
            }
            M3_PROF(7);
            int best_ord = (settled && r == rmax) ? ord : INT_MAX;
            group_butterfly<G>(wl, [&](auto get) { // lexicographic (adj desc, order asc)
                const int a2 = get(adj), o2 = get(best_ord);
                if (a2 > adj || (a2 == adj && o2 < best_ord)) { adj = a2; best_ord = o2; }
            });
            win = __ffsll((long long)ballot_g<G>(settled && ord == best_ord, gl0)) - 1;
        }
    }

    M3_PROF(3);
    // ---- commit (tools.py:3150-3163) -----------------------------------------------------------------
    if (win >= 0) {
        const int Z = __shfl(mp, gl0 + win), stab = __shfl(stab_p, gl0 + win), emp = __shfl(emp_p, gl0 + win);
        const int py = win / W, px = win - py * W;
        res.placed = 1; res.x = px; res.y = py; res.z = Z; res.stab = stab;
        const int cx = tap_div_small(cell, L), cy = cell - cx * L;
        const bool foot = cell < cells && cx >= px && cx < px + bx && cy >= py && cy < py + by;
        // update_level_free_space (:2989-3041) on this cell's column of F, 64 levels at a time.  A cell's new word w
        // depends on its neighbours' OLD word w only, so the column is rewritten word by word -- read, wave-level
        // hand-off, write -- with one word in registers whatever the container's height
        for (int w = 0; w < HW; ++w) {
            u64 nw = 0;
            if (foot) {
                u64 keep = 0;
                if (px > 0 && px + bx < W) {
                    keep = ~0ull;
                    for (int x = px - 1; x <= px + bx; ++x) keep &= ~S.occ[(size_t)(x * L + cy) * HW + w];
                }
                const int lo = Z - 64 * w, hi = Z + bz - 64 * w;        // block = bits [lo, hi) of this word
                const u64 below = lo <= 0 ? 0ull : (lo >= 64 ? ~0ull : ((1ull << lo) - 1ull));
                const u64 upto = hi <= 0 ? 0ull : (hi >= 64 ? ~0ull : ((1ull << hi) - 1ull));
                nw = S.occ[(size_t)cell * HW + w] | (upto & ~below) | (below & ~keep);
            }
            tap_wave_lds_sync(); // every neighbour's word w is read before any is replaced
            if (foot) S.occ[(size_t)cell * HW + w] = nw;
        }
        if (foot) hm = Z + bz;                                                       // :3161
        cnt.valid += vol;
        cnt.empty = emp;
        cnt.nstable += stab;
        if (Z + bz > H) err |= 1;                                                    // level_free_space[zz] IndexError
    }
    cnt.count += 1;
    M3_PROF(4);
#undef M3_PACK
#undef M3_PUSH
#undef M3_EXT_UP
#undef M3_EXT_DOWN
    return res;
}

// One env's whole step as executed by its lane group (stand-alone step in macs.hip, placement waves
// of the fused transition in transition.hip): load state + history, place, store state, feature,
// and -- for the fused form -- the fresh-container start and calc_ratio (flags: TAP_T_*).
template <int G, int WL = 0>
__device__ inline void tap_macs3_wave(const StepArgs &a, int flags, float *ratio_out, int env, int cell, int lane,
                                      int *lds_group)
{
    const int B = a.d.B, W = WL ? WL : a.d.W, Ld = WL ? WL : a.d.L, cells = W * Ld;
    const bool ev = env < B, incell = cell < cells, fresh = flags & TAP_T_FRESH;
    const int gl0 = lane - cell;
    const int HW = macs3_hw(a.d.H);
    const Macs3Lds S = macs3_lds(lds_group, G, a.d.H);

    int hm = 0, cv = 0;
    if (ev && !fresh) {
        if (incell) hm = a.v.hm[(size_t)env * cells + cell];
        if (cell < 4) cv = a.v.cnt[(size_t)env * 4 + cell];
    }
    for (int k = cell; k < G * HW; k += G)           // the env's words are contiguous: coalesced
        S.occ[k] = (ev && !fresh && k < cells * HW) ? a.v.occ[(size_t)env * cells * HW + k] : 0ull;
    Counters cnt = {__shfl(cv, gl0), __shfl(cv, gl0 + 1), __shfl(cv, gl0 + 2), __shfl(cv, gl0 + 3)};
    int bx = 1, by = 1, bz = 1;
    bool act = ev;
    if (ev) {
        if (a.static_) {
            bool badp;
            const long praw = (long)a.ptr[env];
            const long p = tap_col(praw, a.nR, badp);
            const float vx = a.static_[((size_t)env * a.static_rows + 1) * a.nR + p];
            const float vy = a.static_[((size_t)env * a.static_rows + 2) * a.nR + p];
            const float vz = a.static_[((size_t)env * a.static_rows + 3) * a.nR + p];
            bx = badp ? 0 : (int)vx;
            by = badp ? 0 : (int)vy;
            bz = badp ? 0 : (int)vz;
            if (cell == 0) { const float fv[3] = {badp ? 0.f : vx, badp ? 0.f : vy, badp ? 0.f : vz}; tap_step_aux(a, env, 3, fv, praw); }
        } else if (a.blocks_dtype == TAP_DT_F32) {
            const float *b = (const float *)a.blocks + (size_t)env * 3;
            bx = (int)b[0]; by = (int)b[1]; bz = (int)b[2];
        } else {
            const int32_t *b = (const int32_t *)a.blocks + (size_t)env * 3;
            bx = b[0]; by = b[1]; bz = b[2];
        }
        if (a.active) act = a.active[env] != 0;
    }
    int err = 0;
    bool do_step = act;
    if (act && cnt.count >= a.d.n_max) { err |= 2; do_step = false; }
    // sides larger than the container are rejected as invalid input: the reference keeps such a
    // block in its history at (0,0,0) and its later slices run out of range (tools.py:2858, 2914)
    if (act && (bx < 1 || by < 1 || bz < 1 || bx > W || by > Ld)) { err |= 4; do_step = false; }

    S.hm[cell] = hm;
    for (int k = cell; k < 256; k += G) S.lrun[k] = (unsigned char)m3_longest_run((unsigned)k);
    if (ev) // one round trip for the whole placement history: lane i packs entry i
        for (int i = cell; i < cnt.count && i < a.d.n_max; i += G) {
            int f[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) f[q] = (q < 3 ? a.v.pos : a.v.blk)[(size_t)(i * 3 + (q < 3 ? q : q - 3)) * B + env];
            // the placed flag rides on bit 16 of the stored x size
            S.hist[i * MACS3_HIST] = (f[0] & 15) | ((f[1] & 15) << 4) | ((f[3] & 15) << 8) | ((f[4] & 15) << 12) | (((f[3] >> 16) & 1) << 16);
            S.hist[i * MACS3_HIST + 1] = (f[2] & 0xffff) | (f[5] << 16);
        }
    tap_wave_lds_sync();
    const int step = cnt.count;
    const PlaceCfg cfg = {W, Ld, a.d.H, a.d.flags, a.lut};
    const Placement pl = tap_macs3_place<G, WL>(cfg, S, cell, gl0, hm, cnt, err, bx, by, bz, do_step);
    err = group_or<G>(err);

    tap_wave_lds_sync();
    S.hm[cell] = hm;
    tap_wave_lds_sync();
    const int gmax = (flags & TAP_T_RATIO) ? group_max<G>(incell ? hm : 0) : 0;
    if (ev) {
        if (incell && (do_step || fresh)) a.v.hm[(size_t)env * cells + cell] = hm;
        if (do_step || fresh)
            for (int k = cell; k < cells * HW; k += G) a.v.occ[(size_t)env * cells * HW + k] = S.occ[k];
        if (a.feature_out)
            tap_write_feature<3, G>(a.d.feature, W, Ld, S.hm, cell, hm, a.feature_out + (size_t)env * a.flen);
        if (cell == 0) {
            if (do_step || fresh)
                reinterpret_cast<int4 *>(a.v.cnt)[env] = make_int4(cnt.valid, cnt.empty, cnt.nstable, cnt.count);
            if (do_step) {
                a.v.pos[(size_t)(step * 3) * B + env] = pl.x;
                a.v.pos[(size_t)(step * 3 + 1) * B + env] = pl.y;
                a.v.pos[(size_t)(step * 3 + 2) * B + env] = pl.z;
                a.v.stable[(size_t)step * B + env] = (uint8_t)pl.stab;
                a.v.blk[(size_t)(step * 3) * B + env] = bx | (pl.placed << 16); // history of later steps
                a.v.blk[(size_t)(step * 3 + 1) * B + env] = by;                  // (tools.py:2843-2846),
                a.v.blk[(size_t)(step * 3 + 2) * B + env] = bz;                  // failures too
            }
            if (fresh) a.v.err[env] = err;
            else if (err) a.v.err[env] |= err;
            if (flags & TAP_T_RATIO) {                                            // tools.py:3887-3966
                double C = 0.0, P = 0.0, S_ = 0.0;
                if (cnt.count != 0) {
                    C = (double)cnt.valid / (double)((long long)W * Ld * gmax);
                    P = (double)cnt.valid / (double)(cnt.empty + cnt.valid);
                    S_ = (double)cnt.nstable / (double)cnt.count;
                }
                ratio_out[env] = (float)tap_ratio_formula(a.d.ratio_mode, C, P, S_);
            }
        }
    } else if (a.d.feature == TAP_FEAT_ZERO) {
        (void)group_min<G>(INT_MAX);
    }
}
