// rolling.hip -- rolling precedence windows: generate.InitialContainer
// (generate.py:1589-1839) as rolling.validate drives it (rolling.py:589-637), for B instances in
// lock-step.  gfx950 only.
//
// The reference keeps five networkx digraphs over the N blocks of an instance and, per decoding
// step, (1) drops the block the policy picked from the current window, (2) tops the window up to
// `child` nodes with in-degree-0 nodes of what is left of the movement graph, layer by layer in
// node order, (3) cuts the five induced sub-graphs, adding a self-loop to a side graph wherever a
// blocker is still outside the window, and (4) lays them out as the network's static / dynamic
// input.  With N <= 64 every graph is one 64-bit column mask per node ("who blocks me"), the state
// is two masks (entered, window), and one wavefront (lane = node) handles one instance:
// layers are ballots, "lowest k free nodes" is a rank test, sub-matrix entries are bit tests.
//
// One reference quirk is reproduced on purpose: the sub-graph matrices are indexed in the
// iteration order of a CPython set of the window's node ids (networkx FilterAtlas walks
// show_nodes' set when the sub-graph is less than half of the graph), while the static columns use
// the sorted order (generate.py:1758-1761 vs 1774, 1793).  pyset_* below re-states
// Objects/setobject.c for small ints (hash(n) == n): 9 linear probes, perturb shift 5, growth x4
// once fill*5 >= mask*3.
#include <new>

#include "tap_common.h"
#include "tap_masks.h"
#include "tap_place.h"
#include "tap_waves.h"

struct RollArgs {
    int B, D, N, child, W, L, H, arm;
    const int32_t *blocks;    // (B, N, D) rotation 0
    const int32_t *positions; // (B, N, D)  (init only)
    unsigned long long *rel;  // (B, 5, N)  move, left, right, forward, backward column masks
    unsigned long long *state; // (B, 2)    entered, window
    const int64_t *remove_ptr; // (B,) column picked in the previous window, or null
    float *static_out;        // (B, 1+D, child*R)
    float *dynamic_out;       // (B, 3*child, child*R)
    float *colsum_out;        // (B, 3, child*R) nullable
    unsigned long long *bits_out; // (B, child*R) nullable: bit shadow of dynamic_out (tap_dyn_bits layout)
    float *cur_mask_out;      // (B, child*R) nullable
    int32_t *nodes_out;       // (B, child) nullable
    int32_t *err_out;         // (B,) nullable: 1 = window could not be filled
    int err_sticky;           // only ever raise err_out (tap_roller: one flag for the whole episode)
    int wt;                   // flavour of dynamic_out's stores (tap_masks.h: store_stream / tap_write_through)
};

// the window's fp32 tensor is expanded like the step's (tap_masks.h): lane roles rotated by the slab's offset inside
// its 64-byte granule; `dy` = this instance's slab
__device__ __forceinline__ LaneRole roll_lane_role(int v, const float *dy, int C4, int RP, int rows)
{
#ifdef TAP_STREAM_UNALIGNED
    const int add = 0;
#else
    const int add = ((RP * C4) & 3) == 0 ? (int)((reinterpret_cast<uintptr_t>(dy) >> 4) & 3) : 0;
#endif
    return stream_lane_role(v, 0, C4, RP, 65536 / C4 + 1, 0, add, (rows + RP - 1) / RP);
}

__device__ __forceinline__ bool rng_meet(int a0, int a1, int b0, int b1) { return a0 < b1 && b0 < a1; }

// ---- relations: the same box predicates as k_precedence (generate.hip), one lane per node --------
template <int D>
__global__ void __launch_bounds__(TAP_BLOCK) k_rolling_init(RollArgs a)
{
    __shared__ int s_blk[TAP_BLOCK / 64][64 * 3];
    __shared__ int s_pos[TAP_BLOCK / 64][64 * 3];
    const int w = threadIdx.x >> 6, b = threadIdx.x & 63;
    const int inst = blockIdx.x * (TAP_BLOCK / 64) + w;
    const int n = a.N;
    if (inst < a.B)
        for (int k = b; k < n * D; k += 64) {
            s_blk[w][k] = a.blocks[(size_t)inst * n * D + k];
            s_pos[w][k] = a.positions[(size_t)inst * n * D + k];
        }
    tap_wave_lds_sync();
    if (inst >= a.B) return;
    if (b == 0) { a.state[(size_t)inst * 2] = 0; a.state[(size_t)inst * 2 + 1] = 0; }
    if (b >= n) return;
    const int *blk = s_blk[w], *pos = s_pos[w];
#define BX(i) blk[(i) * D]
#define BY(i) (D == 3 ? blk[(i) * D + 1] : 1)
#define BZ(i) blk[(i) * D + D - 1]
#define PX(i) pos[(i) * D]
#define PY(i) (D == 3 ? pos[(i) * D + 1] : 0)
#define PZ(i) pos[(i) * D + D - 1]
    const int x = PX(b), y = PY(b), z = PZ(b), bx = BX(b), by = BY(b), bz = BZ(b);
    const int top = z + bz, z_mid = z + (bz - 1) / 2;
    u64 move = 0, left = 0, right = 0, fwd = 0, bwd = 0;
    if (D == 2) {                                                     // generate.py:575-647
        for (int o = 0; o < n; ++o) {
            if (o == b) continue;
            const int ox = PX(o), oz = PZ(o), obx = BX(o), otop = oz + BZ(o);
            if (rng_meet(ox, ox + obx, x, x + bx) && oz > z) move |= 1ull << o;
            if (x >= a.arm && rng_meet(ox, ox + obx, x - a.arm, x) && otop > z_mid) left |= 1ull << o;
            if (x + bx <= a.W - a.arm && rng_meet(ox, ox + obx, x + bx, x + bx + a.arm) && otop > z_mid) right |= 1ull << o;
        }
        if (x < a.arm) left |= 1ull << b;
        if (x + bx > a.W - a.arm) right |= 1ull << b;
    } else {                                                          // generate.py:649-752
        for (int i = 0; i < bx; ++i)
            for (int j = 0; j < by; ++j) {
                const int cx = x + i, cy = y + j;
                int best = -1, bestz = INT_MAX;
                for (int o = 0; o < n; ++o) {
                    if (o == b) continue;
                    const int ox = PX(o), oy = PY(o), oz = PZ(o);
                    if (cx >= ox && cx < ox + BX(o) && cy >= oy && cy < oy + BY(o) && oz >= top && oz < bestz) { bestz = oz; best = o; }
                }
                if (best >= 0) move |= 1ull << best;
            }
        const int ymid = y + (by - 1) / 2, xmid = x + (bx - 1) / 2;
        for (int o = 0; o < n; ++o) {
            if (o == b) continue;
            const int ox = PX(o), oy = PY(o), otop = PZ(o) + BZ(o);
            if (otop <= z_mid) continue;
            const bool in_y = ymid >= oy && ymid < oy + BY(o), in_x = xmid >= ox && xmid < ox + BX(o);
            if (x > 0 && in_y && x - 1 >= ox && x - 1 < ox + BX(o)) left |= 1ull << o;
            if (x + bx < a.W && in_y && x + bx >= ox && x + bx < ox + BX(o)) right |= 1ull << o;
            if (y > 0 && in_x && y - 1 >= oy && y - 1 < oy + BY(o)) fwd |= 1ull << o;
            if (y + by < a.L && in_x && y + by >= oy && y + by < oy + BY(o)) bwd |= 1ull << o;
        }
        if (x == 0) left |= 1ull << b;
        if (x + bx == a.W) right |= 1ull << b;
        if (y == 0) fwd |= 1ull << b;
        if (y + by == a.L) bwd |= 1ull << b;
    }
    // layout per instance (5*N words): the N movement masks first (every live node's is read each step: 8*N
    // contiguous bytes), then one 4-word record per node with its side masks (only the <= child window nodes'
    // records are read each step: one 32-byte piece of a cache line each instead of four scattered words)
    u64 *r = a.rel + (size_t)inst * 5 * n;
    r[b] = move;
    u64 *q = r + n + 4 * b;
    q[0] = left; q[1] = right; q[2] = fwd; q[3] = bwd;
#undef BX
#undef BY
#undef BZ
#undef PX
#undef PY
#undef PZ
}

// ---- CPython set emulation (see file header) ------------------------------------------------------
__device__ inline int pyset_probe(const int *table, int mask, int key)
{
    unsigned long long perturb = (unsigned long long)key;
    int i = key & mask;
    for (;;) {
        int probes = (i + 9 <= mask) ? 9 : 0, j = i;
        do {
            if (table[j] < 0) return j;
            ++j;
        } while (probes--);
        perturb >>= 5;
        i = (int)((i * 5ull + 1ull + perturb) & (unsigned long long)mask);
    }
}

constexpr int PYSET_CAP = 256;

// keys[0..n) in list order -> order[0..n) in set iteration order; tbl/tmp: PYSET_CAP ints each (n <= 64 keys: the table
// never grows beyond 256 slots whatever the keys' values)
template <class K>
__device__ inline void pyset_order(const K *keys, int n, K *order, int *tbl, int *tmp)
{
    int size = 8, fill = 0;
    for (int i = 0; i < size; ++i) tbl[i] = -1;
    for (int k = 0; k < n; ++k) {
        tbl[pyset_probe(tbl, size - 1, keys[k])] = keys[k];
        ++fill;
        if (fill * 5 >= (size - 1) * 3) {                  // set_table_resize(used * 4)
            int newsize = 8;
            while (newsize <= fill * 4) newsize <<= 1;
            const int oldsize = size;
            for (int i = 0; i < oldsize; ++i) tmp[i] = tbl[i];
            size = newsize;
            for (int i = 0; i < size; ++i) tbl[i] = -1;
            for (int i = 0; i < oldsize; ++i)
                if (tmp[i] >= 0) tbl[pyset_probe(tbl, size - 1, tmp[i])] = tmp[i];
        }
    }
    int m = 0;
    for (int i = 0; i < size; ++i) if (tbl[i] >= 0) order[m++] = (K)tbl[i];
}

// The same for tables of at most 64 slots (n <= 18 keys: 8 -> 32 slots), held across the wave: lane l
// owns slot l, occupancy is one 64-bit mask, a probe is bit arithmetic on it and a re-insert reads the
// key with a lane broadcast -- no memory, no dependent LDS chain.  Every lane of the wave must call.
__device__ __forceinline__ int pyset_probe_mask(u64 occ, int mask, int key)
{
    unsigned long long perturb = (unsigned long long)key;
    int i = key & mask;
    for (;;) {
        const int probes = (i + 9 <= mask) ? 9 : 0;
        const u64 span = (2ull << probes) - 1ull, free_ = ~(occ >> i) & span;
        if (free_) return i + __ffsll((long long)free_) - 1;
        perturb >>= 5;
        i = (int)((i * 5ull + 1ull + perturb) & (unsigned long long)mask);
    }
}

__device__ inline void pyset_order_wave(const unsigned char *keys, int n, unsigned char *order, int lane)
{
    const int mykey = lane < n ? (int)keys[lane] : 0;   // one LDS read; the loops then read lanes, not memory
    // OR over lanes 0..17 (n <= 18: one 16-lane row and two lanes of the next) without a loop over the keys:
    // DPP steps inside the rows, then the two rows' first lanes
    auto or_keys = [&](unsigned v) -> unsigned {
        v |= (unsigned)tap_dpp<0xB1>((int)v);
        v |= (unsigned)tap_dpp<0x4E>((int)v);
        v |= (unsigned)tap_dpp<0x141>((int)v);
        v |= (unsigned)tap_dpp<0x140>((int)v);
        return (unsigned)__builtin_amdgcn_readlane((int)v, 0) | (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    };
    const int fsize = n < 5 ? 8 : 32;   // n <= 18: the table grows once, 8 -> 32 slots, at the fifth key
    const int home = mykey & (fsize - 1);
    {
        // Closed form.  Small ints hash to themselves; a key whose home slot (key & mask) is free goes
        // there, and a key is only ever displaced by a key that shares its home.  So when all keys
        // have distinct homes in the FINAL table (its size depends on n alone), every key sits at its
        // home whatever the insertion and resize history was, and the iteration order is the order of
        // the homes.  Only sets with two keys congruent modulo the table size need more.
        const unsigned homes = or_keys(lane < n ? 1u << home : 0u);
        if (__popc(homes) == n) {
            if (lane < n) order[__popc(homes & ((1u << home) - 1u))] = (unsigned char)mykey;
            return;
        }
    }
    if (n >= 5) {
        // Half-closed form.  The first five keys go through the 8-slot table and are re-inserted, in ITS slot
        // order, into the empty 32-slot table: if their homes there are distinct they end at home whatever that
        // order was, and only the remaining keys need the probe sequence, in list order.
        const unsigned h5 = or_keys(lane < 5 ? 1u << home : 0u);
        if (__popc(h5) == 5) {
            int val = -1;
            unsigned occ = h5;                           // 32 slots: 32-bit occupancy
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int key = __builtin_amdgcn_readlane(mykey, k);
                if (lane == (key & 31)) val = key;
            }
            for (int k = 5; k < n; ++k) {
                const int key = __builtin_amdgcn_readlane(mykey, k);
                int slot = key & 31;
                if ((occ >> slot) & 1u) slot = pyset_probe_mask((u64)occ, 31, key);   // home taken: the probe sequence
                if (lane == slot) val = key;
                occ |= 1u << slot;
            }
            if (lane < 32 && ((occ >> lane) & 1u)) order[__popc(occ & ((1u << lane) - 1u))] = (unsigned char)val;
            return;
        }
    }
    int val = -1, size = 8, fill = 0;
    u64 occ = 0;
    for (int k = 0; k < n; ++k) {
        // the table state is wave-uniform: say so, and the probe arithmetic runs on the scalar unit
        const int key = __builtin_amdgcn_readlane(mykey, k);
        const int slot = pyset_probe_mask(occ, size - 1, key);
        if (lane == slot) val = key;
        occ |= 1ull << slot;
        ++fill;
        if (fill * 5 >= (size - 1) * 3) {                  // set_table_resize(used * 4)
            int newsize = 8;
            while (newsize <= fill * 4) newsize <<= 1;
            u64 old = occ;
            const int oldval = val;
            val = -1; occ = 0; size = newsize;
            while (old) {                                  // re-insert in slot order
                const int s0 = __ffsll((long long)old) - 1;
                old &= old - 1ull;
                const int key2 = __builtin_amdgcn_readlane(oldval, s0);
                const int slot2 = pyset_probe_mask(occ, size - 1, key2);
                if (lane == slot2) val = key2;
                occ |= 1ull << slot2;
            }
        }
    }
    if ((occ >> lane) & 1ull) order[__popcll(occ & ((1ull << lane) - 1ull))] = (unsigned char)val;
}

// ---- one window step: remove, top up, cut sub-graphs, emit tensors ----------------------------------
struct RollFastLds {      // the compile-time-shaped emission (rolling_emit_fast)
    unsigned iw[64];       // lane (k, cm): relation k of sub-graph column cm as a word over sub-graph rows
    unsigned cw32[64];     // dynamic's column words (3*CH <= 32 rows)
    float lut[16][4];      // nibble -> four 0.f / 1.f
    int dims[16][4];       // sorted slot -> the node's block sides
};
struct alignas(16) RollLds { // one wavefront's scratch: 4.8 KB, so that 8 workgroups (all 32 waves a CU
                          // gets at B = 8192) fit the 160 KB of LDS
    unsigned char lst[64]; // sub_graph_nodes in list order
    unsigned char ord[64]; // sub-graph node order (matrix index -> node)
    unsigned char pos[64]; // node -> matrix index
    unsigned char srt[64]; // sorted position -> node
    union {
        int tbl[2 * PYSET_CAP]; // set-order emulation (step 3)
        u64 cw[PYSET_CAP];      // later: dynamic's column words, bit (sec*child + rm) of cw[col]
        RollFastLds f;
    };
    u64 side[5][64];      // column masks by sub-graph index (rolling_emit_fast: by node id)
};

// one wavefront = one instance, lane v = node v; every lane of the wave must call this
// -DTAP_PROF: per-phase shader-clock deltas of the first 8192 instances (scratch/prof_roll.py reads them)
#ifdef TAP_PROF
__device__ unsigned int tap_prof_w[8192 * 8];
extern "C" int tap_prof_read(unsigned int *out)
{
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(tap_prof_w), sizeof(unsigned int) * 8192 * 8);
    return 0;
}
#define PROF(i) do { const long long t_ = clock64(); if (v == 0 && inst < 8192) tap_prof_w[inst * 8 + (i)] = (unsigned)(t_ - tp); tp = t_; } while (0)
#define PROF_BEGIN long long tp = clock64()
#elif defined(TAP_ROLL_STOP)
// -DTAP_ROLL_STOP: the wave returns at phase mark tap_roll_stop_at (scratch/valu_phases.py counts SQ_INSTS_VALU per
// phase by differencing); results are garbage past the mark, the state must be restored by the caller
__constant__ int tap_roll_stop_at = -1;
extern "C" int tap_prof_set_stop(int i) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(tap_roll_stop_at), &i, sizeof(int)); }
#define PROF(i) do { if (tap_roll_stop_at == (i)) return; } while (0)
#define PROF_BEGIN do { } while (0)
#else
#define PROF(i) do { } while (0)
#define PROF_BEGIN do { } while (0)
#endif

// Per-node inputs of the tensor emission: the four side masks and the block sizes, only for window nodes
struct RollNode {
    u64 rel[5];
    int bdim[3];
};

template <int D>
__device__ __forceinline__ RollNode rolling_node_loads(const RollArgs &a, int inst, int v, bool on, u64 rel0)
{
    RollNode nd = {{rel0, 0, 0, 0, 0}, {0, 0, 0}};
    if (on) {
#pragma unroll
        for (int k = 1; k < 5; ++k) nd.rel[k] = a.rel[(size_t)inst * 5 * a.N + a.N + 4 * v + (k - 1)];   // the node's record
#pragma unroll
        for (int k = 0; k < D; ++k) nd.bdim[k] = a.blocks[((size_t)inst * a.N + v) * D + k];
    }
    return nd;
}

// (4) tensors (generate.py:1778-1822) of one instance by one wavefront, lane v = node v: S.ord holds the
//     sub-graph node order, (entered, window) the state after the graph step.  LDS = any struct with ord, pos,
//     srt (bytes), cw (u64 per column, packed form only) and side[5][>= child] (u64).
template <int D, class LDS>
__device__ inline void rolling_emit_wave(const RollArgs &a, int inst, int v, LDS &S, u64 entered, u64 window,
                                         const RollNode &nd)
{
    const int N = a.N, child = a.child;
    constexpr int R = D == 2 ? 2 : 6;
    const int nRc = child * R;
    const u64 all = N == 64 ? ~0ull : ((1ull << N) - 1ull);
    const u64 bit = 1ull << v, below = bit - 1ull;
    const bool inwin = (window & bit) != 0;
    const u64 (&rel)[5] = nd.rel;
    const int (&bdim)[3] = nd.bdim;
    PROF_BEGIN;
    if (v < child) S.pos[S.ord[v]] = (unsigned char)v;
    tap_wave_lds_sync();

    // (4) tensors (generate.py:1778-1822).  Window-node lanes publish their five column masks (with
    //     the :1690-1705 rule: a blocker that has not entered any window yet => the side counts as
    //     self-blocked) by sub-graph index.
    const u64 after = all & ~entered;               // after_nodes_list
    if (inwin) {
        const int midx = S.pos[v];
        S.side[0][midx] = rel[0] & window;
#pragma unroll
        for (int k = 1; k < 5; ++k) S.side[k][midx] = (rel[k] & window) | ((rel[k] & after) ? bit : 0ull);
        S.srt[__popcll(window & below)] = (unsigned char)v;     // sorted position -> node (static's column order)
        if (a.nodes_out) a.nodes_out[(size_t)inst * child + __popcll(window & below)] = v;
    }
    tap_wave_lds_sync();
    PROF(4);
    
    const int perm2[2][3] = {{0, 1, 0}, {1, 0, 0}};
    const int perm3[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    float *st = a.static_out + (size_t)inst * (1 + D) * nRc;
    float *dy = a.dynamic_out ? a.dynamic_out + (size_t)inst * 3 * child * nRc : nullptr;   // NULL: no fp32 expansion (tapenv.h)
    // which side pair guards rotation r (:1808-1821): the axis that becomes vertical
    auto side_of = [&](int r, int sec) -> int {
        const int *p = D == 2 ? perm2[r] : perm3[r];
        if (sec == 0) return 0;
        if (p[D - 1] == 0) return sec;              // left / right
        if (D == 3 && p[D - 1] == 1) return 2 + sec; // forward / backward
        return -1;                                   // up / down: empty
    };
    const int rows = 3 * child;
    const bool packed = rows <= 64 && (nRc & 3) == 0 && nRc <= 256;
    // The five masks of a window node re-indexed from node ids to sub-graph rows (bit rm = node ord[rm]) are
    // the same for every rotation's column, so they are gathered ONCE per (relation, node) -- lane = (k, cm),
    // 5 * child <= 64 lanes when child <= 12 -- instead of three times in each of the child*R column lanes.
    const bool once = packed && 5 * child <= 64 && child <= 21;
    unsigned *iw = reinterpret_cast<unsigned *>(&S.side[0][0]);  // re-uses the relation slots (hand-off below)
    if (once) {
        const int k5 = v / child, cm5 = v - k5 * child;
        const bool on5 = v < 5 * child;
        const u64 m = on5 ? S.side[k5][cm5] : 0ull;
        const int ordv = S.ord[v < child ? v : 0];  // lane rm holds ord[rm]: one LDS read, then lane reads
        unsigned w = 0u;
        for (int rm = 0; rm < child; ++rm) {
            const int node = __builtin_amdgcn_readlane(ordv, rm);                // wave-uniform
            const unsigned half = node >= 32 ? (unsigned)(m >> 32) : (unsigned)m;
            w |= ((half >> (node & 31)) & 1u) << rm;
        }
        tap_wave_lds_sync();                        // every lane holds its mask: the slots may be overwritten
        if (on5) iw[v] = w;
        tap_wave_lds_sync();
    }
    PROF(5);
    // lane = column (r, cm): static, column sums, initial mask, and the column's word of `dynamic`.  When the
    // tensor fits the bit-shadow form (3*child <= 64 rows, nRc % 4 == 0, nRc <= 256) the words go through LDS
    // and the fp32 tensor is expanded from them as in stream_wave_bits (tap_masks.h): a store instruction
    // covers 64/(nRc/4) whole rows with 16 bytes per lane, nontemporal.  Otherwise each lane walks its column.
    for (int col0 = 0; col0 < nRc; col0 += 64) {
        const int col = col0 + v;
        const bool oncol = col < nRc;
        const int r = oncol ? col / child : 0, cm = oncol ? col - r * child : 0;
        const int *p = D == 2 ? perm2[r] : perm3[r];
        // static: cm = sorted slot; the node's sides come from its own lane (loaded with `rel`)
        const int node_s = S.srt[cm];
        int side_len[3];
#pragma unroll
        for (int k = 0; k < D; ++k) side_len[k] = __shfl(bdim[k], node_s);
        if (!oncol) continue;
        st[col] = (float)cm;                                                      // :1795-1801
        for (int k = 0; k < D; ++k)
            st[(size_t)(1 + k) * nRc + col] = (float)(p[k] == 0 ? side_len[0] : p[k] == 1 ? side_len[1] : side_len[2]);
        if (once) {
            unsigned ws[3];
#pragma unroll
            for (int sec = 0; sec < 3; ++sec) {
                const int k = side_of(r, sec);
                ws[sec] = k < 0 ? 0u : iw[k * child + cm];                        // dynamic: cm = sub-graph index
                if (a.colsum_out) a.colsum_out[((size_t)inst * 3 + sec) * nRc + col] = (float)__popc(ws[sec]);
            }
            if (a.cur_mask_out)                                                   // model.py:297-307
                a.cur_mask_out[(size_t)inst * nRc + col] = (__popc(ws[1]) * __popc(ws[2]) + __popc(ws[0]) != 0) ? 0.f : 1.f;
            const u64 w = (u64)ws[0] | ((u64)ws[1] << child) | ((u64)ws[2] << (2 * child));
            S.cw[col] = w;
            if (a.bits_out) a.bits_out[(size_t)inst * nRc + col] = w;
            continue;
        }
        u64 m[3];
        float sum[3];
#pragma unroll
        for (int sec = 0; sec < 3; ++sec) {
            const int k = side_of(r, sec);
            m[sec] = k < 0 ? 0ull : S.side[k][cm];                             // dynamic: cm = sub-graph index
            sum[sec] = (float)__popcll(m[sec]);
            if (a.colsum_out) a.colsum_out[((size_t)inst * 3 + sec) * nRc + col] = sum[sec];
        }
        if (a.cur_mask_out)                                                       // model.py:297-307
            a.cur_mask_out[(size_t)inst * nRc + col] = (sum[1] * sum[2] + sum[0] != 0.f) ? 0.f : 1.f;
        if (packed) {
            unsigned ws[3] = {0u, 0u, 0u};               // child <= 21 bits per section: 32-bit arithmetic
            for (int rm = 0; rm < child; ++rm) {
                const int node = __builtin_amdgcn_readfirstlane((int)S.ord[rm]), sh = node & 31; // wave-uniform
                const bool hi = node >= 32;
#pragma unroll
                for (int sec = 0; sec < 3; ++sec)
                    ws[sec] |= (((hi ? (unsigned)(m[sec] >> 32) : (unsigned)m[sec]) >> sh) & 1u) << rm;
            }
            const u64 w = (u64)ws[0] | ((u64)ws[1] << child) | ((u64)ws[2] << (2 * child));
            S.cw[col] = w;
            if (a.bits_out) a.bits_out[(size_t)inst * nRc + col] = w;
        } else {
            for (int rm = 0; rm < child; ++rm) {
                const int node = S.ord[rm];
#pragma unroll
                for (int sec = 0; sec < 3; ++sec)
                    if (dy) dy[(size_t)(sec * child + rm) * nRc + col] = (float)((m[sec] >> node) & 1ull);
            }
        }
    }
    
    if (!packed) return;
    tap_wave_lds_sync();
    PROF(6);
    const int C4 = nRc >> 2, RP = 64 / C4;
    if (dy && v < RP * C4) {
        // lane roles rotated so that every store instruction starts on a 64-byte granule (tap_masks.h: stream_lane_role)
        const LaneRole role = roll_lane_role(v, dy, C4, RP, rows);
        const int c4 = role.c4;
        const u64 w0 = S.cw[c4 * 4], w1 = S.cw[c4 * 4 + 1], w2 = S.cw[c4 * 4 + 2], w3 = S.cw[c4 * 4 + 3];
        float4 *dst = reinterpret_cast<float4 *>(dy) + c4;
        int rw = role.r0;
        for (int i = 0; i < role.nq; ++i, rw += RP) {
            if ((unsigned)rw >= (unsigned)rows) continue;
            store_stream(&dst[(size_t)rw * C4], make_float4(bit_as_float(w0, rw), bit_as_float(w1, rw), bit_as_float(w2, rw),
                                                            bit_as_float(w3, rw)), a.wt);
        }
    }
    PROF(7);
}

// (4) again, for the window sizes known at compile time (CH nodes: 3*CH <= 32 rows, CH*R <= 64 columns, 5*CH <= 64
//     (relation, column) pairs): the same tensors with every index computation, loop bound and shift a constant.
//     Differences from rolling_emit_wave that matter for the instruction count (SQ_INSTS_VALU per window wave at c5:
//     emission 450 -> ~150, profiles/r03_*):
//       * window lanes publish their RAW masks by node id; the (relation, column) lanes mask them, test the
//         "blocker outside every window so far" rule (:1690-1705) once per mask instead of once per node lane, and
//         pick the CH row bits with the node ids in scalar registers;
//       * the block sides go through LDS rows by sorted slot, so a column lane reads the side its rotation puts in
//         a row with one address computation instead of three cross-lane reads and a select chain;
//       * the fp32 expansion is a table look-up: lane (rsub, c4) interleaves the bits of its four column words
//         once (rows rsub + RP*q sit RP >= 4 bits apart, so the four columns' bits of one row form a nibble),
//         and a row's float4 is lut[nibble] -- one bit-field extract and one shift per 16-byte store.
// the compile-time-shaped second half of rolling_emit_fast: S.f.iw (the 5*CH row words), S.f.dims (block sides by
// sorted slot), S.f.lut and S.srt are in place; columns, masks, static and the float4 expansion of `dynamic`.
// (also the tail of the two-word form for 10-node windows, rolling_window_wave2)
template <int D, int CH>
__device__ __forceinline__ void rolling_emit_fast_tail(const RollArgs &a, int inst, int v, RollLds &S)
{
    constexpr int R = D == 2 ? 2 : 6, NRC = CH * R, C4 = NRC / 4, RP = 64 / C4, ROWS = 3 * CH, QN = (ROWS + RP - 1) / RP;
#if defined(TAP_PROF)
    long long tp = clock64();
#endif
    float *st = a.static_out + (size_t)inst * (1 + D) * NRC;
    float *dy = a.dynamic_out ? a.dynamic_out + (size_t)inst * ROWS * NRC : nullptr;   // NULL: no fp32 expansion (tapenv.h)
    if (v < NRC) {
        const int col = v, r = v / CH, cm = v - r * CH;
        // which side pair guards rotation r (:1808-1821): the axis that becomes vertical, p[D-1] of the rotation's
        // permutation -- 3D: (2,1,2,0,1,0)[r], two bits each; 2D: (1,0)[r]
        const int pz = D == 3 ? (0x126 >> (2 * r)) & 3 : 1 - r;
        const bool guarded = pz == 0 || (D == 3 && pz == 1);
        const int k1 = pz == 0 ? 1 : 3;                                  // left/right or forward/backward
        const unsigned ws0 = S.f.iw[cm];
        const unsigned r1 = S.f.iw[k1 * CH + cm], r2 = S.f.iw[(k1 + 1) * CH + cm];
        const unsigned ws1 = guarded ? r1 : 0u, ws2 = guarded ? r2 : 0u;
        const int c0 = __popc(ws0), c1 = __popc(ws1), c2 = __popc(ws2);
        if (a.colsum_out) {
            a.colsum_out[((size_t)inst * 3 + 0) * NRC + col] = (float)c0;
            a.colsum_out[((size_t)inst * 3 + 1) * NRC + col] = (float)c1;
            a.colsum_out[((size_t)inst * 3 + 2) * NRC + col] = (float)c2;
        }
        if (a.cur_mask_out) a.cur_mask_out[(size_t)inst * NRC + col] = (c1 * c2 + c0 != 0) ? 0.f : 1.f;   // model.py:297-307
        const unsigned w = ws0 | (ws1 << CH) | (ws2 << (2 * CH));
        S.f.cw32[col] = w;
        if (a.bits_out) a.bits_out[(size_t)inst * NRC + col] = (u64)w;
        // static (:1795-1801): row 0 = sorted slot, row 1+k = side perm[r][k] of the node in that slot; the three
        // sides' positions in the slot's LDS row as nibbles (4 * index), one constant per row
        st[col] = (float)cm;
        const int r4 = 4 * r;
        const int o0 = D == 3 ? (0x884400 >> r4) & 15 : (0x40 >> r4) & 15;   // 3D: (0,0,1,1,2,2)[r]   2D: (0,1)[r]
        const int o1 = D == 3 ? (0x408084 >> r4) & 15 : (0x04 >> r4) & 15;   // 3D: (1,2,0,2,0,1)[r]   2D: (1,0)[r]
        const char *row = reinterpret_cast<const char *>(&S.f.dims[cm][0]);
        st[(size_t)1 * NRC + col] = (float)*reinterpret_cast<const int *>(row + o0);
        st[(size_t)2 * NRC + col] = (float)*reinterpret_cast<const int *>(row + o1);
        if (D == 3) {
            const int o2 = (0x040848 >> r4) & 15;                             //     (2,1,2,0,1,0)[r]
            st[(size_t)3 * NRC + col] = (float)*reinterpret_cast<const int *>(row + o2);
        }
    }
    tap_wave_lds_sync();
    PROF(6);
    {
        // Lane roles rotated by sb = (slab start / 16) mod 4 so that every store instruction starts on a 64-byte
        // granule (tap_masks.h: stream_lane_role): a 7 200-byte slab (c5) starts 32 bytes into one for every other
        // instance.  The sb lanes that wrapped run one instruction late.
        constexpr int K = RP * C4;
        constexpr bool AL = (K & 3) == 0;
#ifdef TAP_STREAM_UNALIGNED
        const int sb = 0;
#else
        const int sb = AL ? __builtin_amdgcn_readfirstlane((int)((reinterpret_cast<uintptr_t>(dy) >> 4) & 3)) : 0;
#endif
        if (dy && v < K) {
            int role = v - sb;
            const int late = role < 0 ? 1 : 0;
            role += late ? K : 0;
            const int rsub = role / C4, c4 = role - rsub * C4;
            const uint4 cq = *reinterpret_cast<const uint4 *>(&S.f.cw32[c4 * 4]);
            unsigned mk = 0u;
#pragma unroll
            for (int q = 0; q < QN; ++q) mk |= 1u << (RP * q);
            // bit RP*q + i of `il` = column 4*c4 + i at row rsub + RP*q
            unsigned il = (cq.x >> rsub) & mk;
            il |= ((cq.y >> rsub) & mk) << 1;
            il |= ((cq.z >> rsub) & mk) << 2;
            il |= ((cq.w >> rsub) & mk) << 3;
            float4 *dst = reinterpret_cast<float4 *>(dy) + c4 + (size_t)rsub * C4;
            // the late lanes play roles K - sb .. K - 1 (row groups (K - 3) / C4 and up): one more instruction only if
            // such a lane still has a row in its last round
            constexpr int NI = QN + ((AL && RP * (QN - 1) + (K - 3) / C4 < ROWS) ? 1 : 0);
            int q = -late;
#pragma unroll
            for (int i = 0; i < NI; ++i, ++q) {
                if (q >= 0 && RP * q + rsub < ROWS) {
                    const unsigned nib = (il >> (RP * q)) & 15u;
                    store_stream(&dst[(size_t)RP * q * C4], *reinterpret_cast<const float4 *>(&S.f.lut[nib][0]), a.wt);
                }
            }
        }
    }
    PROF(7);
}

template <int D, int CH>
__device__ __forceinline__ void rolling_emit_fast(const RollArgs &a, int inst, int v, RollLds &S, u64 entered, u64 window,
                                                  const RollNode &nd)
{
    constexpr int R = D == 2 ? 2 : 6, NRC = CH * R, C4 = NRC / 4, RP = 64 / C4, ROWS = 3 * CH;
    static_assert(ROWS <= 32 && NRC <= 64 && NRC % 4 == 0 && 5 * CH <= 64 && RP >= 4 && CH <= 16, "shape outside the fast emission");
    const int N = a.N;
    const u64 all = N == 64 ? ~0ull : ((1ull << N) - 1ull);
    const u64 after = all & ~entered;                                  // after_nodes_list
    const bool inwin = ((window >> v) & 1ull) != 0;
    PROF_BEGIN;
    if (inwin) {
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(window >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)window, 0u));
#pragma unroll
        for (int k = 0; k < 5; ++k) S.side[k][v] = nd.rel[k];
        S.srt[rank] = (unsigned char)v;                                // sorted position -> node (static's column order)
        *reinterpret_cast<int4 *>(&S.f.dims[rank][0]) = make_int4(nd.bdim[0], nd.bdim[1], nd.bdim[2], 0);
        if (a.nodes_out) a.nodes_out[(size_t)inst * CH + rank] = v;
    }
    if (v < 16)
        *reinterpret_cast<float4 *>(&S.f.lut[v][0]) = make_float4((float)(v & 1), (float)((v >> 1) & 1), (float)((v >> 2) & 1), (float)((v >> 3) & 1));
    tap_wave_lds_sync();
    PROF(4);
    // lane (k5, cm5): relation k5 of the node in sub-graph column cm5, re-indexed to sub-graph rows
    {
        const int k5 = v / CH, cm5 = v - k5 * CH;
        const bool on5 = v < 5 * CH;
        const int t = S.ord[on5 ? cm5 : 0];
        const u64 m = S.side[on5 ? k5 : 0][t];
        const unsigned mlo = (unsigned)m, mhi = (unsigned)(m >> 32);
        unsigned ow[(CH + 3) / 4];                                      // the sub-graph order, four node ids per scalar
#pragma unroll
        for (int j = 0; j < (CH + 3) / 4; ++j)
            ow[j] = (unsigned)__builtin_amdgcn_readfirstlane((int)reinterpret_cast<const unsigned *>(S.ord)[j]);
        unsigned w = 0u;
#pragma unroll
        for (int rm = 0; rm < CH; ++rm) {
            const unsigned node = (ow[rm >> 2] >> (8 * (rm & 3))) & 0xffu;        // wave-uniform
            const unsigned half = node >= 32u ? mhi : mlo;
            w |= ((half >> (node & 31u)) & 1u) << rm;
        }
        // :1690-1705 a side blocker that has not entered any window yet => the node counts as blocking itself
        const bool self = k5 >= 1 && (m & after) != 0ull;
        w |= self ? (1u << cm5) : 0u;
        if (on5) S.f.iw[v] = w;
    }
    tap_wave_lds_sync();
    PROF(5);
    rolling_emit_fast_tail<D, CH>(a, inst, v, S);
}


// ---- 65 .. 256 blocks per instance: still ONE wavefront per instance, lane v = nodes v, v + 64, v + 128, v + 192 ----
// Every graph is NW = ceil(N / 64) 64-bit words per node (the layout k_rolling_init_big writes: the N movement masks
// first, then per node its four side masks), the state is (entered, window) of NW words each, held wave-uniform.  The
// steps are the ones above: a layer is NW ballots, a node's rank counts the words below its own.  The emission turns
// every window node's masks into words over the sub-graph ROWS first (bit rm = node ord[rm]), after which nothing
// depends on N any more.  Windows of at most 64 / NW nodes (side[k][NW * slot + word] fills the relation slots
// exactly: 32 nodes up to 128 blocks, 21 up to 192, 16 up to 256); the tensor goes out packed (rolling_emit_wave's
// float4 expansion) when it has the bit-shadow shape, element by element otherwise.  (Round 4 had this form for two
// words only; 129 .. 256 blocks ran one THREAD per instance: 853 against 80 us per step at N = 130 / 128.)
constexpr int ROLL_CH_WIDE = -2;      // template tags: this form with NW = -CH words instead of the one-word graph step
// window sizes with a compile-time-shaped kernel (0 = any window, shapes read from the arguments)
__host__ __device__ constexpr bool roll_fast_ok(int D, int child) { return child == 10 && (D == 2 || D == 3); }

template <int D, int NW>
__device__ inline void rolling_window_waveN(const RollArgs &a, int inst, int v, RollLds &S)
{
    static_assert(NW >= 2 && NW <= 4, "two to four words per node mask");
    if (inst >= a.B) return;
    const int N = a.N, child = a.child;
    constexpr int R = D == 2 ? 2 : 6;
    const int nRc = child * R;
    const u64 bit = 1ull << v, below = bit - 1ull;
    u64 all[NW];
    bool has[NW];                                                     // this lane's h-th node exists
#pragma unroll
    for (int h = 0; h < NW; ++h) {
        all[h] = N >= 64 * (h + 1) ? ~0ull : (N > 64 * h ? ((1ull << (N - 64 * h)) - 1ull) : 0ull);
        has[h] = v + 64 * h < N;
    }
    auto uniform64 = [](u64 x) -> u64 {
        return ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) |
               (unsigned)__builtin_amdgcn_readfirstlane((int)x);
    };
    const u64 *relb = a.rel + (size_t)inst * 5 * N * NW;
    u64 m[NW][NW];                                                    // m[h] = movement mask of node v + 64 h
#pragma unroll
    for (int h = 0; h < NW; ++h) {
        const u64 *q = relb + (size_t)(has[h] ? v + 64 * h : 0) * NW;
        if constexpr (NW % 2 == 0) {
#pragma unroll
            for (int k = 0; k < NW; k += 2) {
                const ulonglong2 t = *reinterpret_cast<const ulonglong2 *>(q + k);
                m[h][k] = has[h] ? t.x : 0ull; m[h][k + 1] = has[h] ? t.y : 0ull;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NW; ++k) { const u64 t = q[k]; m[h][k] = has[h] ? t : 0ull; }
        }
    }
    u64 *stp = a.state + (size_t)inst * 2 * NW;
    u64 sraw[2 * NW];
#pragma unroll
    for (int k = 0; k < 2 * NW; ++k) sraw[k] = stp[k];
    const long ptr_raw = a.remove_ptr ? (long)a.remove_ptr[inst] : 0;
    u64 e[NW], w[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) { e[k] = uniform64(sraw[k]); w[k] = uniform64(sraw[NW + k]); }

    // (1) remove_block(sub_graph_nodes[ptr mod child])  rolling.py:632-637, generate.py:1824-1835
    if (a.remove_ptr) {
        const long slot = tap_mod_col((long)uniform64((u64)ptr_raw), child, nRc);
        int base = 0;
        u64 hitm[NW];
#pragma unroll
        for (int h = 0; h < NW; ++h) {
            const bool hit = (w[h] & bit) && base + __popcll(w[h] & below) == slot;
            hitm[h] = __ballot(hit);
            base += __popcll(w[h]);
        }
#pragma unroll
        for (int h = 0; h < NW; ++h) w[h] &= ~hitm[h];
    }
    // (2) top the window up (generate.py:1724-1750)
    int count = 0;
#pragma unroll
    for (int h = 0; h < NW; ++h) {
        if (w[h] & bit) S.lst[count + __popcll(w[h] & below)] = (unsigned char)(v + 64 * h);
        count += __popcll(w[h]);
    }
    u64 ad[NW];
#pragma unroll
    for (int h = 0; h < NW; ++h) ad[h] = 0ull;
    while (count < child) {
        u64 g[NW];                                                    // nodes still in gm_copy
        int left = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) { g[k] = all[k] & ~(e[k] | ad[k]); left += __popcll(g[k]); }
        const bool single = left == 1;
        bool f[NW];
        u64 fm[NW], any = 0ull;
#pragma unroll
        for (int h = 0; h < NW; ++h) {
            u64 blocked = 0ull;
#pragma unroll
            for (int k = 0; k < NW; ++k) blocked |= m[h][k] & g[k];
            f[h] = has[h] && (g[h] & bit) && (single || blocked == 0ull);
            fm[h] = __ballot(f[h]);
            any |= fm[h];
        }
        if (any == 0ull) break;
        const int need = child - count;
        int base = 0;
        u64 tm[NW];
#pragma unroll
        for (int h = 0; h < NW; ++h) {
            const int r = base + __popcll(fm[h] & below);
            const bool t = f[h] && r < need;
            if (t) S.lst[count + r] = (unsigned char)(v + 64 * h);
            tm[h] = __ballot(t);
            base += __popcll(fm[h]);
        }
#pragma unroll
        for (int h = 0; h < NW; ++h) { ad[h] |= tm[h]; count += __popcll(tm[h]); }
    }
#pragma unroll
    for (int h = 0; h < NW; ++h) { e[h] |= ad[h]; w[h] |= ad[h]; }
    const int short_window = count != child;
    // the side records and block sides of this lane's window node, in flight under the set order.  A lane seldom holds
    // two window nodes (v, v + 64, ...): the others' are fetched afterwards, when they exist (keeps the registers free)
    int h1 = -1, nin = 0;                                             // the first half this lane has a window node in
#pragma unroll
    for (int h = NW - 1; h >= 0; --h) if (w[h] & bit) { h1 = h; ++nin; }
    u64 sd[4][NW];
    int bd[3] = {0, 0, 0};
    {
        const bool on = h1 >= 0 && !short_window;
        const int node = v + 64 * (h1 < 0 ? 0 : h1);
        const u64 *q = relb + (size_t)N * NW + (size_t)(on ? node : 0) * 4 * NW;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int x = 0; x < NW; ++x) { const u64 t = q[k * NW + x]; sd[k][x] = on ? t : 0ull; }
#pragma unroll
        for (int k = 0; k < D; ++k) bd[k] = on ? a.blocks[((size_t)inst * N + node) * D + k] : 0;
    }
    tap_wave_lds_sync();

    // (3) node order of the induced sub-graphs: 2 * child <= 64 < N, always the set order (:1684-1688, 1758-1761)
    if (!short_window) {
        if (child <= 18) pyset_order_wave(S.lst, child, S.ord, v);
        else if (v == 0) pyset_order(S.lst, child, S.ord, S.tbl, S.tbl + PYSET_CAP);
    }
    tap_wave_lds_sync();
#ifndef TAP_ROLL_LATEWAIT
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) while only loads are outstanding (see rolling_window_wave)
#endif
    if (v == 0) {
#pragma unroll
        for (int k = 0; k < NW; ++k) { stp[k] = e[k]; stp[NW + k] = w[k]; }
        if (a.err_out && (short_window || !a.err_sticky)) a.err_out[inst] = short_window;
    }
    if (short_window) return;

    // (4) tensors (generate.py:1778-1822).  Scratch inside the (now idle) set-order tables: cw (packed form, at most
    //     126 columns) | iw: 5*child row words | dims: the block sides by sorted slot
    //     10-node windows (the reference's --nodes 10) finish in rolling_emit_fast's compile-time-shaped tail, which
    //     reads the row words, the block sides and the nibble table from RollFastLds.
    const bool fast = roll_fast_ok(D, child);
    unsigned *iw = fast ? S.f.iw : reinterpret_cast<unsigned *>(S.tbl) + 256;
    int *dims = fast ? &S.f.dims[0][0] : S.tbl + 416;
    const int dstride = fast ? 4 : 3;
    if (fast && v < 16)
        *reinterpret_cast<float4 *>(&S.f.lut[v][0]) = make_float4((float)(v & 1), (float)((v >> 1) & 1), (float)((v >> 2) & 1), (float)((v >> 3) & 1));
    u64 *side0 = &S.side[0][0];                                        // [5][64] words: relation k, slot, word
    auto slot_of = [&](int h) -> int {                                 // sorted position of node v + 64 h
        int sl = __popcll(w[h] & below);
#pragma unroll
        for (int k = 0; k < NW; ++k) if (k < h) sl += __popcll(w[k]);
        return sl;
    };
    auto publish = [&](int h, const u64 (&mv)[NW], const u64 (&rec)[4][NW], const int (&sides)[3]) {
        const int node = v + 64 * h;
        const int slot = slot_of(h);
#pragma unroll
        for (int x = 0; x < NW; ++x) side0[NW * slot + x] = mv[x];
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int x = 0; x < NW; ++x) side0[(1 + k) * 64 + NW * slot + x] = rec[k][x];
#pragma unroll
        for (int k = 0; k < 3; ++k) dims[slot * dstride + k] = sides[k];
        S.srt[slot] = (unsigned char)node;
        if (a.nodes_out) a.nodes_out[(size_t)inst * child + slot] = node;
    };
#pragma unroll
    for (int h = 0; h < NW; ++h) if (h == h1) publish(h, m[h], sd, bd);
    if (__ballot(nin > 1)) {                                           // rare: a lane with two or more window nodes
#pragma unroll
        for (int h = 1; h < NW; ++h) {
            if (nin > 1 && h != h1 && (w[h] & bit)) {
                const u64 *q = relb + (size_t)N * NW + (size_t)(v + 64 * h) * 4 * NW;
                u64 rec[4][NW];
                int sides[3] = {0, 0, 0};
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int x = 0; x < NW; ++x) rec[k][x] = q[k * NW + x];
#pragma unroll
                for (int k = 0; k < D; ++k) sides[k] = a.blocks[((size_t)inst * N + v + 64 * h) * D + k];
                publish(h, m[h], rec, sides);
            }
        }
    }
    tap_wave_lds_sync();
    u64 af[NW];                                                        // after_nodes_list
#pragma unroll
    for (int k = 0; k < NW; ++k) af[k] = all[k] & ~e[k];
    for (int idx = v; idx < 5 * child; idx += 64) {                    // (relation k, sub-graph column cm)
        const int k = idx / child, cm = idx - k * child;
        const int u = S.ord[cm], uh = u >> 6;
        const u64 ubelow = (1ull << (u & 63)) - 1ull;
        int slot = 0;
#pragma unroll
        for (int x = 0; x < NW; ++x) slot += x < uh ? __popcll(w[x]) : x == uh ? __popcll(w[x] & ubelow) : 0;
        u64 wd[NW];
        u64 outside = 0ull;
#pragma unroll
        for (int x = 0; x < NW; ++x) { wd[x] = side0[k * 64 + NW * slot + x]; outside |= wd[x] & af[x]; }
        // :1690-1705: a blocker that has not entered any window yet => the side counts as self-blocked
        const bool selfb = k > 0 && outside != 0ull;
#pragma unroll
        for (int x = 0; x < NW; ++x) { wd[x] &= w[x]; if (selfb && x == uh) wd[x] |= 1ull << (u & 63); }
        unsigned ww = 0u;
        for (int rm = 0; rm < child; ++rm) {
            const int node = S.ord[rm], nh = node >> 6;
            u64 sel = wd[0];
#pragma unroll
            for (int x = 1; x < NW; ++x) sel = nh == x ? wd[x] : sel;
            ww |= (unsigned)((sel >> (node & 63)) & 1ull) << rm;
        }
        iw[idx] = ww;
    }
    tap_wave_lds_sync();
    if (fast) {                                                        // kernel-uniform
        rolling_emit_fast_tail<D, 10>(a, inst, v, S);
        return;
    }
    const int perm2[2][3] = {{0, 1, 0}, {1, 0, 0}};
    const int perm3[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    float *st = a.static_out + (size_t)inst * (1 + D) * nRc;
    float *dy = a.dynamic_out ? a.dynamic_out + (size_t)inst * 3 * child * nRc : nullptr;   // NULL: no fp32 expansion (tapenv.h)
    const int rows = 3 * child;
    const bool packed = rows <= 64 && (nRc & 3) == 0 && nRc <= 128;
    for (int col = v; col < nRc; col += 64) {
        const int r = col / child, cm = col - r * child;
        const int *p = D == 2 ? perm2[r] : perm3[r];
        st[col] = (float)cm;                                                          // :1795-1801, cm = sorted slot
        for (int k = 0; k < D; ++k) st[(size_t)(1 + k) * nRc + col] = (float)dims[cm * dstride + p[k]];
        unsigned ws[3];
#pragma unroll
        for (int sec = 0; sec < 3; ++sec) {                                           // :1808-1821, cm = sub-graph index
            int k = 0;
            if (sec > 0) k = p[D - 1] == 0 ? sec : (D == 3 && p[D - 1] == 1) ? 2 + sec : -1;
            ws[sec] = k < 0 ? 0u : iw[k * child + cm];
            if (a.colsum_out) a.colsum_out[((size_t)inst * 3 + sec) * nRc + col] = (float)__popc(ws[sec]);
        }
        if (a.cur_mask_out)                                                           // model.py:297-307
            a.cur_mask_out[(size_t)inst * nRc + col] = (__popc(ws[1]) * __popc(ws[2]) + __popc(ws[0]) != 0) ? 0.f : 1.f;
        if (packed) {
            const u64 cwv = (u64)ws[0] | ((u64)ws[1] << child) | ((u64)ws[2] << (2 * child));
            S.cw[col] = cwv;
            if (a.bits_out) a.bits_out[(size_t)inst * nRc + col] = cwv;
        } else {
            for (int rm = 0; rm < child; ++rm)
#pragma unroll
                for (int sec = 0; sec < 3; ++sec)
                    if (dy) dy[(size_t)(sec * child + rm) * nRc + col] = (float)((ws[sec] >> rm) & 1u);
        }
    }
    if (!packed) return;
    tap_wave_lds_sync();
    const int C4 = nRc >> 2, RP = 64 / C4;
    if (dy && v < RP * C4) {
        const LaneRole role = roll_lane_role(v, dy, C4, RP, rows);       // store instructions start on 64-byte granules
        const int c4 = role.c4;
        const u64 c0 = S.cw[c4 * 4], c1 = S.cw[c4 * 4 + 1], c2 = S.cw[c4 * 4 + 2], c3 = S.cw[c4 * 4 + 3];
        float4 *dst = reinterpret_cast<float4 *>(dy) + c4;
        int rw = role.r0;
        for (int i = 0; i < role.nq; ++i, rw += RP) {
            if ((unsigned)rw >= (unsigned)rows) continue;
            store_stream(&dst[(size_t)rw * C4], make_float4(bit_as_float(c0, rw), bit_as_float(c1, rw), bit_as_float(c2, rw),
                                                            bit_as_float(c3, rw)), a.wt);
        }
    }
}

template <int D, int CH>
__device__ inline void rolling_window_wave(const RollArgs &a, int inst, int v, RollLds &S)
{
    if constexpr (CH <= ROLL_CH_WIDE) { rolling_window_waveN<D, -CH>(a, inst, v, S); return; }
    if (inst >= a.B) return;
    PROF_BEGIN;
    const int N = a.N, child = CH > 0 ? CH : a.child;
    constexpr int R = D == 2 ? 2 : 6;
    const int nRc = child * R;
    const u64 all = N == 64 ? ~0ull : ((1ull << N) - 1ull);
    const u64 bit = 1ull << v, below = bit - 1ull;
    const bool isnode = v < N;
    // Every input of the graph step is requested up front: the window state, the previous pick and each
    // node's movement mask.  The four side masks and the block sizes are only needed for the (at most
    // `child`) nodes of the new window and are requested once it is known, under the set-order emulation.
    auto uniform64 = [](u64 x) -> u64 {
        return ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(x >> 32)) << 32) |
               (unsigned)__builtin_amdgcn_readfirstlane((int)x);
    };
    const u64 rel0 = isnode ? a.rel[(size_t)inst * 5 * N + v] : 0ull;
    const u64 st_entered = a.state[(size_t)inst * 2], st_window = a.state[(size_t)inst * 2 + 1];
    const long ptr_raw = a.remove_ptr ? (long)a.remove_ptr[inst] : 0;
    // (entered, window) are wave-uniform: in scalar registers the set arithmetic below costs no VALU slots
    u64 entered = uniform64(st_entered), window = uniform64(st_window);

    PROF(0);
    // (1) remove_block(sub_graph_nodes[ptr mod child])  rolling.py:632-637, generate.py:1824-1835
    if (a.remove_ptr) {
        long slot = (long)uniform64((u64)ptr_raw);
        slot = tap_mod_col(slot, child, nRc);                        // rolling.py:632-633
        const bool hit = (window & bit) && __popcll(window & below) == slot;
        window &= ~__ballot(hit);
    }
    // (2) top the window up: in-degree-0 nodes of gm_copy, layer by layer, ascending ids
    //     (generate.py:1724-1750); list order = old window (sorted by the previous call) + appended
    int count = __popcll(window);
    if (window & bit) S.lst[__popcll(window & below)] = (unsigned char)v;
    u64 added = 0;
    while (count < child) {
        const u64 gmc = all & ~(entered | added);                    // nodes still in gm_copy
        const bool free_ = isnode && (gmc & bit) && (__popcll(gmc) == 1 || (rel0 & gmc) == 0);
        const u64 fm = __ballot(free_);
        if (fm == 0) break;
        const int need = child - count;
        const bool take = free_ && __popcll(fm & below) < need;
        if (take) S.lst[count + __popcll(fm & below)] = (unsigned char)v;
        const u64 tm = __ballot(take);
        added |= tm;
        count += __popcll(tm);
    }
    entered |= added;   // after_nodes_list.remove (:1745) and, the window being full, decompose() (:1712-1721)
    window |= added;
    const int short_window = count != child;
    const bool inwin = (window & bit) != 0;
    // in flight under the set order (asking for every node's record up front instead -- one round trip fewer on the
    // wave's critical path, 44 more bytes per node -- measured 17.6 against 15.8 us per fused step at c5)
    RollNode nd = rolling_node_loads<D>(a, inst, v, inwin && !short_window, rel0);
    tap_wave_lds_sync();

    PROF(1);
    // (3) node order of the induced sub-graphs (:1684-1688, 1758-1761)
    if (2 * child < N) {
        if (child <= 18) { if (!short_window) pyset_order_wave(S.lst, child, S.ord, v); } // wave-uniform branch
        else if (v == 0 && !short_window) pyset_order(S.lst, child, S.ord, S.tbl, S.tbl + PYSET_CAP);
    } else if (inwin) {
        S.ord[__popcll(window & below)] = (unsigned char)v;
    }
    tap_wave_lds_sync();
    // The node records requested above are consumed by the emission, AFTER the first stores below: on gfx9 one counter
    // (vmcnt) covers loads and stores, so a wait placed at the records' first use would also wait for every store issued
    // before it to be ACKNOWLEDGED (the round-3 ISA had five such vmcnt(0) between the state store and the fp32
    // expansion).  Waiting here, while only loads are outstanding -- they have had the whole set order to arrive --
    // leaves nothing for the stores to be waited on.
#ifndef TAP_ROLL_LATEWAIT                                             // (A/B builds: the round-3 placement of the waits)
    __builtin_amdgcn_s_waitcnt(0x0F70);                              // vmcnt(0)
#endif
    if (v == 0) {
        a.state[(size_t)inst * 2] = entered;
        a.state[(size_t)inst * 2 + 1] = window;
        if (a.err_out && (short_window || !a.err_sticky)) a.err_out[inst] = short_window;
    }
    if (short_window) return;
    PROF(2);
    if constexpr (CH > 0) rolling_emit_fast<D, CH>(a, inst, v, S, entered, window, nd);
    else rolling_emit_wave<D>(a, inst, v, S, entered, window, nd);
    PROF(3);
}

// instances the two-word wavefront form takes (rolling_window_wave2)
// words per node mask of the one-wavefront form for instances above 64 blocks (0: no such form -- one thread per instance)
__host__ __device__ constexpr int roll_wide_nw(int N, int child) { return (N > 64 && N <= 256 && child * ((N + 63) / 64) <= 64) ? (N + 63) / 64 : 0; }
__host__ __device__ constexpr bool roll_wide_ok(int N, int child) { return roll_wide_nw(N, child) != 0; }

// 8 waves per SIMD (<= 64 VGPRs): at B = 8192 a CU gets 32 one-wave instances, and at 68 VGPRs only 28
// were resident, so one workgroup in eight ran as a second round
// The leading scalar arguments of the window kernels repeat what a window wave needs to ADDRESS its first loads (relation
// masks, window state, the previous pick): they are preloaded into SGPRs by the dispatcher (-mllvm
// -amdgpu-kernarg-preload-count, set for this file by the Makefile), so those loads do not wait for a read of the
// argument block through the scalar cache (as in transition.hip).
#define ROLL_HOT_PARAMS unsigned long long *h_rel, unsigned long long *h_state, const int64_t *h_remove, int h_B, int h_N
#define ROLL_HOT_ARGS(r) (r).rel, (r).state, (r).remove_ptr, (r).B, (r).N
__device__ __forceinline__ RollArgs roll_hot(const RollArgs &k, ROLL_HOT_PARAMS)
{
    RollArgs r = k;
    r.rel = h_rel; r.state = h_state; r.remove_ptr = h_remove; r.B = h_B; r.N = h_N;
    return r;
}

template <int D, int CH>
__global__ void __launch_bounds__(TAP_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) k_rolling_window(ROLL_HOT_PARAMS, RollArgs a)
{
    __shared__ RollLds S[TAP_BLOCK / 64];
    const int w = TAP_WAVE_INDEX(), v = threadIdx.x & 63;
    rolling_window_wave<D, CH>(roll_hot(a, h_rel, h_state, h_remove, h_B, h_N), blockIdx.x * (TAP_BLOCK / 64) + w, v, S[w]);
}

// the three- and four-word forms (129 .. 256 blocks) keep up to 16 mask words per lane: no 64-register cap
template <int D, int NW>
__global__ void __launch_bounds__(TAP_BLOCK) k_rolling_window_wide(ROLL_HOT_PARAMS, RollArgs a)
{
    __shared__ RollLds S[TAP_BLOCK / 64];
    const int w = TAP_WAVE_INDEX(), v = threadIdx.x & 63;
    rolling_window_waveN<D, NW>(roll_hot(a, h_rel, h_state, h_remove, h_B, h_N), blockIdx.x * (TAP_BLOCK / 64) + w, v, S[w]);
}

// ---- fused rolling step: add_new_block for the column picked in the CURRENT window (gathered from
//      its static tensor) and the NEXT window, in one launch.  Both depend only on that pick, so the
//      placement waves (lane-per-cell groups, tap_waves.h) and the window waves (one per instance)
//      of a workgroup run side by side, as in tap_transition.
struct RollStepArgs {
    RollArgs r;
    StepArgs s;
};

#ifndef TAP_ROLL_EPB
#define TAP_ROLL_EPB 2      // A/B builds (-DTAP_ROLL_EPB=1 | 4).  Round 6, six processes per build, c5 M env-steps/s: 1: 426-478, 2: 542-558, 4: 483-533
#endif
constexpr int ROLL_EPB = TAP_ROLL_EPB;   // instances per workgroup of the fused step (2: 3-wave workgroups pack a CU's 28 wave slots
                              // better than 6-wave ones: 9 x 3 = 27 against 4 x 6 = 24)

template <int D, int G, bool SOFT, int CH>
__device__ __forceinline__ void rolling_step_body(const RollStepArgs &a, ROLL_HOT_PARAMS);

template <int D, int G, int CH>
__global__ void __launch_bounds__((64 * (ROLL_EPB + ROLL_EPB * G / 64 + (ROLL_EPB * G % 64 ? 1 : 0)))) __attribute__((amdgpu_waves_per_eu(8, 8)))
k_rolling_step_soft(ROLL_HOT_PARAMS, RollStepArgs a)
{
    rolling_step_body<D, G, true, CH>(a, h_rel, h_state, h_remove, h_B, h_N);
}

template <int D, int G, int CH>
__global__ void __launch_bounds__((64 * (ROLL_EPB + ROLL_EPB * G / 64 + (ROLL_EPB * G % 64 ? 1 : 0)))) __attribute__((amdgpu_waves_per_eu(7, 8)))
k_rolling_step(ROLL_HOT_PARAMS, RollStepArgs a)
{
    rolling_step_body<D, G, false, CH>(a, h_rel, h_state, h_remove, h_B, h_N);
}

template <int D, int G, bool SOFT, int CH>
__device__ __forceinline__ void rolling_step_body(const RollStepArgs &a, ROLL_HOT_PARAMS)
{
    constexpr int EPB = ROLL_EPB;                           // instances per workgroup
    constexpr int ENV_WAVES = (EPB * G + 63) / 64;
    __shared__ RollLds S[EPB];
    __shared__ int s_old[64 * ENV_WAVES];
    __shared__ int s_new[64 * ENV_WAVES];
    // The wave's index in a scalar register (as in transition.hip): the window wave's instance, and with it every address
    // it loads from and stores to, becomes scalar arithmetic.  c5 differs by +- 5 % from process to process (buffer
    // placement), so this A/B took eight processes per build in one session (profiles/r06_rolling_swave_ab.txt): median
    // 507 -> 541 M env-steps/s, best 534 -> 557 M.  -DTAP_ROLL_VWAVE: the index as a vector value (A/B builds).
#ifdef TAP_ROLL_VWAVE
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
#else
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#endif
    const int base = blockIdx.x * EPB;
    if (wave < ENV_WAVES) {
        __builtin_amdgcn_s_setprio(2);
        const int cell = tid % G, grp = tid / G;
        // groups beyond EPB (when EPB*G is not a multiple of 64) idle on an out-of-range env
        const int env = grp < EPB ? base + grp : a.s.d.B;
        tap_lb_place_wave<D, G, !SOFT>(a.s, 0, nullptr, env, cell, lane, s_old + (tid - cell), s_new + (tid - cell));
        return;
    }
    rolling_window_wave<D, CH>(roll_hot(a.r, h_rel, h_state, h_remove, h_B, h_N), base + (wave - ENV_WAVES), lane, S[wave - ENV_WAVES]);
}

// ---- more than 64 blocks per instance ------------------------------------------------------------------------
// The kernels above hold a node per lane and a graph per 64-bit mask.  Instances of 65 .. 256 blocks (the
// reference's --total_blocks_num is free, rolling.py:702) take this path instead: the same steps with masks of
// NW = ceil(N/64) words, ONE THREAD per instance (relations: one thread per node) -- a correctness path like
// big.hip, not a tuned one.  Layout: rel = 5*N*NW words per instance (N movement masks, then per node the four
// side masks), state = 2*NW words (entered, window).
// (Round 5: instances of up to ROLL_MAX_N = 4 096 blocks -- the masks are compiled for 4, 16 or 64 words, the smallest
//  that holds N; above 256 blocks this is the only form.)
constexpr int ROLL_MAX_N = 4096;

template <int MW> struct NMaskT {
    u64 w[MW];
};
template <int MW> __device__ __forceinline__ bool nm_test(const NMaskT<MW> &m, int i) { return (m.w[i >> 6] >> (i & 63)) & 1ull; }
template <int MW> __device__ __forceinline__ void nm_set(NMaskT<MW> &m, int i) { m.w[i >> 6] |= 1ull << (i & 63); }
template <int MW> __device__ __forceinline__ void nm_clear(NMaskT<MW> &m, int i) { m.w[i >> 6] &= ~(1ull << (i & 63)); }
template <int MW> __device__ __forceinline__ int nm_count(const NMaskT<MW> &m, int NW) { int c = 0; for (int k = 0; k < NW; ++k) c += __popcll(m.w[k]); return c; }
template <int MW> __device__ __forceinline__ bool nm_meets(const u64 *a, const NMaskT<MW> &b, int NW) { bool r = false; for (int k = 0; k < NW; ++k) r |= (a[k] & b.w[k]) != 0; return r; }

template <int D, int MW>
__global__ void __launch_bounds__(TAP_BLOCK) k_rolling_init_big(RollArgs a)
{
    typedef NMaskT<MW> NMask;
    const long t = (long)blockIdx.x * TAP_BLOCK + threadIdx.x;
    const int n = a.N, NW = (n + 63) / 64;
    if (t >= (long)a.B * n) return;
    const int inst = (int)(t / n), b = (int)(t - (long)inst * n);
    const int32_t *blk = a.blocks + (size_t)inst * n * D, *pos = a.positions + (size_t)inst * n * D;
    if (b == 0) for (int k = 0; k < 2 * NW; ++k) a.state[(size_t)inst * 2 * NW + k] = 0;
#define BX(i) blk[(i) * D]
#define BY(i) (D == 3 ? blk[(i) * D + 1] : 1)
#define BZ(i) blk[(i) * D + D - 1]
#define PX(i) pos[(i) * D]
#define PY(i) (D == 3 ? pos[(i) * D + 1] : 0)
#define PZ(i) pos[(i) * D + D - 1]
    const int x = PX(b), y = PY(b), z = PZ(b), bx = BX(b), by = BY(b), bz = BZ(b);
    const int top = z + bz, z_mid = z + (bz - 1) / 2;
    NMask m[5] = {};
    if (D == 2) {                                                     // generate.py:575-647
        for (int o = 0; o < n; ++o) {
            if (o == b) continue;
            const int ox = PX(o), oz = PZ(o), obx = BX(o), otop = oz + BZ(o);
            if (rng_meet(ox, ox + obx, x, x + bx) && oz > z) nm_set(m[0], o);
            if (x >= a.arm && rng_meet(ox, ox + obx, x - a.arm, x) && otop > z_mid) nm_set(m[1], o);
            if (x + bx <= a.W - a.arm && rng_meet(ox, ox + obx, x + bx, x + bx + a.arm) && otop > z_mid) nm_set(m[2], o);
        }
        if (x < a.arm) nm_set(m[1], b);
        if (x + bx > a.W - a.arm) nm_set(m[2], b);
    } else {                                                          // generate.py:649-752
        for (int i = 0; i < bx; ++i)
            for (int j = 0; j < by; ++j) {
                const int cx = x + i, cy = y + j;
                int best = -1, bestz = INT_MAX;
                for (int o = 0; o < n; ++o) {
                    if (o == b) continue;
                    const int ox = PX(o), oy = PY(o), oz = PZ(o);
                    if (cx >= ox && cx < ox + BX(o) && cy >= oy && cy < oy + BY(o) && oz >= top && oz < bestz) { bestz = oz; best = o; }
                }
                if (best >= 0) nm_set(m[0], best);
            }
        const int ymid = y + (by - 1) / 2, xmid = x + (bx - 1) / 2;
        for (int o = 0; o < n; ++o) {
            if (o == b) continue;
            const int ox = PX(o), oy = PY(o), otop = PZ(o) + BZ(o);
            if (otop <= z_mid) continue;
            const bool in_y = ymid >= oy && ymid < oy + BY(o), in_x = xmid >= ox && xmid < ox + BX(o);
            if (x > 0 && in_y && x - 1 >= ox && x - 1 < ox + BX(o)) nm_set(m[1], o);
            if (x + bx < a.W && in_y && x + bx >= ox && x + bx < ox + BX(o)) nm_set(m[2], o);
            if (y > 0 && in_x && y - 1 >= oy && y - 1 < oy + BY(o)) nm_set(m[3], o);
            if (y + by < a.L && in_x && y + by >= oy && y + by < oy + BY(o)) nm_set(m[4], o);
        }
        if (x == 0) nm_set(m[1], b);
        if (x + bx == a.W) nm_set(m[2], b);
        if (y == 0) nm_set(m[3], b);
        if (y + by == a.L) nm_set(m[4], b);
    }
#undef BX
#undef BY
#undef BZ
#undef PX
#undef PY
#undef PZ
    u64 *r = a.rel + (size_t)inst * 5 * n * NW;
    for (int k = 0; k < NW; ++k) r[(size_t)b * NW + k] = m[0].w[k];
    u64 *q = r + (size_t)n * NW + (size_t)b * 4 * NW;
    for (int s5 = 0; s5 < 4; ++s5) for (int k = 0; k < NW; ++k) q[s5 * NW + k] = m[1 + s5].w[k];
}

// remove_block + convert_to_input of ONE instance by one thread (rolling_window_wave, serial form)
template <int D, int MW>
__global__ void __launch_bounds__(64) k_rolling_window_big(RollArgs a)
{
    typedef NMaskT<MW> NMask;
    typedef unsigned short node_t;                                    // node ids up to 4 095
    const int inst = blockIdx.x * 64 + threadIdx.x;
    if (inst >= a.B) return;
    const int N = a.N, NW = (N + 63) / 64, child = a.child;
    constexpr int R = D == 2 ? 2 : 6;
    const int nRc = child * R;
    const u64 *rel0 = a.rel + (size_t)inst * 5 * N * NW;
    const u64 *side = rel0 + (size_t)N * NW;
    u64 *stp = a.state + (size_t)inst * 2 * NW;
    NMask entered = {}, window = {}, all = {};
    for (int k = 0; k < NW; ++k) { entered.w[k] = stp[k]; window.w[k] = stp[NW + k]; }
    for (int i = 0; i < N; ++i) nm_set(all, i);
    if (a.remove_ptr) {                                               // (1) rolling.py:632-637
        long slot = tap_mod_col((long)a.remove_ptr[inst], child, nRc);
        for (int i = 0; i < N; ++i)
            if (nm_test(window, i) && slot-- == 0) { nm_clear(window, i); break; }
    }
    node_t lst[80], ord[80];
    int count = 0;                                                    // (2) generate.py:1724-1750
    for (int i = 0; i < N; ++i) if (nm_test(window, i)) lst[count++] = (node_t)i;
    NMask added = {};
    while (count < child) {
        NMask gmc;
        for (int k = 0; k < NW; ++k) gmc.w[k] = all.w[k] & ~(entered.w[k] | added.w[k]);
        const bool single = nm_count(gmc, NW) == 1;
        const int need = child - count;
        int got = 0;
        NMask take = {};
        for (int j = 0; j < N && got < need; ++j)
            if (nm_test(gmc, j) && (single || !nm_meets(rel0 + (size_t)j * NW, gmc, NW))) { nm_set(take, j); lst[count + got++] = (node_t)j; }
        if (got == 0) break;
        for (int k = 0; k < NW; ++k) added.w[k] |= take.w[k];
        count += got;
    }
    for (int k = 0; k < NW; ++k) { entered.w[k] |= added.w[k]; window.w[k] |= added.w[k]; }
    const int short_window = count != child;
    for (int k = 0; k < NW; ++k) { stp[k] = entered.w[k]; stp[NW + k] = window.w[k]; }
    if (a.err_out && (short_window || !a.err_sticky)) a.err_out[inst] = short_window;
    if (short_window) return;
    int tbl[PYSET_CAP], tmp[PYSET_CAP];                               // the set-order tables, thread-private
    if (2 * child < N) pyset_order(lst, child, ord, tbl, tmp);        // (3)
    else { int m = 0; for (int i = 0; i < N; ++i) if (nm_test(window, i)) ord[m++] = (node_t)i; }
    // (4) tensors (generate.py:1778-1822)
    NMask after;
    for (int k = 0; k < NW; ++k) after.w[k] = all.w[k] & ~entered.w[k];
    const int perm2[2][3] = {{0, 1, 0}, {1, 0, 0}};
    const int perm3[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
    float *st = a.static_out + (size_t)inst * (1 + D) * nRc;
    float *dy = a.dynamic_out ? a.dynamic_out + (size_t)inst * 3 * child * nRc : nullptr;   // NULL: no fp32 expansion (tapenv.h)
    const int rows = 3 * child;
    int srt_i = 0;
    for (int v = 0; v < N; ++v) {                                     // sorted position -> node: static's columns
        if (!nm_test(window, v)) continue;
        if (a.nodes_out) a.nodes_out[(size_t)inst * child + srt_i] = v;
        for (int r = 0; r < R; ++r) {
            const int *p = D == 2 ? perm2[r] : perm3[r];
            const int col = r * child + srt_i;
            st[col] = (float)srt_i;
            for (int k = 0; k < D; ++k) st[(size_t)(1 + k) * nRc + col] = (float)a.blocks[((size_t)inst * N + v) * D + p[k]];
        }
        ++srt_i;
    }
    for (int cm = 0; cm < child; ++cm) {                              // dynamic: cm = sub-graph index
        const int v = ord[cm];
        for (int r = 0; r < R; ++r) {
            const int *p = D == 2 ? perm2[r] : perm3[r];
            const int col = r * child + cm;
            int sum[3] = {0, 0, 0};
            u64 word = 0;
            for (int sec = 0; sec < 3; ++sec) {
                int k = 0;                                            // which relation guards this rotation (:1808-1821)
                if (sec > 0) k = p[D - 1] == 0 ? sec : (D == 3 && p[D - 1] == 1) ? 2 + sec : -1;
                for (int rm = 0; rm < child; ++rm) {
                    const int u = ord[rm];
                    bool bit = false;
                    if (k == 0) bit = nm_test(window, u) && ((rel0[(size_t)v * NW + (u >> 6)] >> (u & 63)) & 1ull);
                    else if (k > 0) {
                        const u64 *sm = side + ((size_t)v * 4 + (k - 1)) * NW;
                        bit = (sm[u >> 6] >> (u & 63)) & 1ull;        // u is a window node
                        if (u == v && !bit) {                         // :1690-1705 a blocker outside every window so far
                            bool out = false;
                            for (int q = 0; q < NW; ++q) out |= (sm[q] & after.w[q]) != 0;
                            bit = out;
                        }
                    }
                    if (dy) dy[(size_t)(sec * child + rm) * nRc + col] = bit ? 1.f : 0.f;
                    sum[sec] += bit;
                    if (bit && sec * child + rm < 64) word |= 1ull << (sec * child + rm);
                }
                if (a.colsum_out) a.colsum_out[((size_t)inst * 3 + sec) * nRc + col] = (float)sum[sec];
            }
            if (a.cur_mask_out) a.cur_mask_out[(size_t)inst * nRc + col] = (sum[1] * sum[2] + sum[0] != 0) ? 0.f : 1.f;
            if (a.bits_out && rows <= 64) a.bits_out[(size_t)inst * nRc + col] = word;
        }
    }
}

static int roll_check(tap_ctx *ctx, int B, int D, int N, int child)
{
    // windows: at most 64 nodes (one 64-bit word of sub-graph rows per node); instances of 65 .. 256 blocks keep the
    // one-wavefront kernels while the window has at most 64 / ceil(N / 64) nodes (roll_wide_nw), beyond that -- and above
    // 256 blocks -- one thread per instance
    if ((D != 2 && D != 3) || B < 0 || N < 1 || N > ROLL_MAX_N || child < 1 || child > N || child > 64)
        return tap_fail(ctx, TAP_E_INVALID, "bad rolling arguments (total blocks <= %d, window <= min(total, 64); instances above 64 "
                                            "blocks run on one wavefront each up to 256 blocks with windows of at most 64 / ceil(N / 64) "
                                            "nodes, one thread per instance otherwise)", ROLL_MAX_N);
    return TAP_OK;
}

extern "C" int tap_rolling_init(tap_ctx *ctx, int B, int D, int N, const int32_t *container_size,
                                int arm_size, const int32_t *blocks, const int32_t *positions,
                                uint64_t *rel_out, uint64_t *state_out, void *stream)
{
    int rc = roll_check(ctx, B, D, N, 1);
    if (rc) return rc;
    if (!container_size || !blocks || !positions || !rel_out || !state_out || arm_size < 1)
        return tap_fail(ctx, TAP_E_INVALID, "null pointer");
    RollArgs a = {};
    a.B = B; a.D = D; a.N = N; a.W = container_size[0]; a.L = D == 3 ? container_size[1] : 1;
    a.H = container_size[D - 1]; a.arm = arm_size; a.blocks = blocks; a.positions = positions;
    a.rel = reinterpret_cast<unsigned long long *>(rel_out);
    a.state = reinterpret_cast<unsigned long long *>(state_out);
    const int grid = (B + TAP_BLOCK / 64 - 1) / (TAP_BLOCK / 64);
    if (grid == 0) return TAP_OK;
    if (N > 64) {                                                     // one thread per node, multi-word masks
        const long threads = (long)B * N;
        const unsigned g2 = (unsigned)((threads + TAP_BLOCK - 1) / TAP_BLOCK);
        hipStream_t s_ = (hipStream_t)stream;
#define TAP_ROLL_INIT(D_) do { \
            if (N <= 256) hipLaunchKernelGGL((k_rolling_init_big<D_, 4>), dim3(g2), dim3(TAP_BLOCK), 0, s_, a); \
            else if (N <= 1024) hipLaunchKernelGGL((k_rolling_init_big<D_, 16>), dim3(g2), dim3(TAP_BLOCK), 0, s_, a); \
            else hipLaunchKernelGGL((k_rolling_init_big<D_, 64>), dim3(g2), dim3(TAP_BLOCK), 0, s_, a); } while (0)
        if (D == 2) TAP_ROLL_INIT(2); else TAP_ROLL_INIT(3);
#undef TAP_ROLL_INIT
        TAP_LAUNCH_CHECK(ctx, "k_rolling_init_big");
        return TAP_OK;
    }
    if (D == 2) hipLaunchKernelGGL(k_rolling_init<2>, dim3(grid), dim3(TAP_BLOCK), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_rolling_init<3>, dim3(grid), dim3(TAP_BLOCK), 0, (hipStream_t)stream, a);
    TAP_LAUNCH_CHECK(ctx, "k_rolling_init");
    return TAP_OK;
}

static int rolling_window_impl(tap_ctx *ctx, int B, int D, int N, int child, const int32_t *blocks,
                               const uint64_t *rel, uint64_t *state, const int64_t *remove_ptr,
                               float *static_out, float *dynamic_out, float *colsum_out,
                               uint64_t *bits_out, float *current_mask_out, int32_t *nodes_out,
                               int32_t *err_out, void *stream, int err_sticky);

extern "C" int tap_rolling_window(tap_ctx *ctx, int B, int D, int N, int child, const int32_t *blocks,
                                  const uint64_t *rel, uint64_t *state, const int64_t *remove_ptr,
                                  float *static_out, float *dynamic_out, float *colsum_out,
                                  uint64_t *bits_out, float *current_mask_out, int32_t *nodes_out,
                                  int32_t *err_out, void *stream)
{
    return rolling_window_impl(ctx, B, D, N, child, blocks, rel, state, remove_ptr, static_out, dynamic_out, colsum_out,
                               bits_out, current_mask_out, nodes_out, err_out, stream, 0);
}

static int rolling_window_impl(tap_ctx *ctx, int B, int D, int N, int child, const int32_t *blocks,
                               const uint64_t *rel, uint64_t *state, const int64_t *remove_ptr,
                               float *static_out, float *dynamic_out, float *colsum_out,
                               uint64_t *bits_out, float *current_mask_out, int32_t *nodes_out,
                               int32_t *err_out, void *stream, int err_sticky)
{
    int rc = roll_check(ctx, B, D, N, child);
    if (rc) return rc;
    if (!blocks || !rel || !state || !static_out) return tap_fail(ctx, TAP_E_INVALID, "null pointer");
    if (!dynamic_out && !bits_out)
        return tap_fail(ctx, TAP_E_INVALID, "dynamic_out may only be left out when bits_out takes the window's bit shadow");
    RollArgs a = {};
    a.err_sticky = err_sticky;
    a.wt = tap_write_through((size_t)B * 3 * child * child * (D == 2 ? 2 : 6) * sizeof(float));
    a.B = B; a.D = D; a.N = N; a.child = child; a.blocks = blocks;
    a.rel = reinterpret_cast<unsigned long long *>(const_cast<uint64_t *>(rel));
    a.state = reinterpret_cast<unsigned long long *>(state);
    a.remove_ptr = remove_ptr; a.static_out = static_out; a.dynamic_out = dynamic_out;
    a.colsum_out = colsum_out; a.cur_mask_out = current_mask_out; a.nodes_out = nodes_out; a.err_out = err_out;
    a.bits_out = reinterpret_cast<unsigned long long *>(bits_out);
    if (bits_out && (3 * child > 64 || (child * (D == 2 ? 2 : 6)) % 4 != 0 || child * (D == 2 ? 2 : 6) > 256))
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "bits_out needs 3*child <= 64, (child*R) %% 4 == 0, child*R <= 256");
    const int grid = (B + TAP_BLOCK / 64 - 1) / (TAP_BLOCK / 64);
    if (grid == 0) return TAP_OK;
    if (const int nw = roll_wide_nw(N, child)) {                      // 65 .. 256 blocks: one wavefront per instance
        hipStream_t s_ = (hipStream_t)stream;
#define TAP_ROLL_WIDE(D_) do { \
            if (nw == 2) hipLaunchKernelGGL((k_rolling_window<D_, ROLL_CH_WIDE>), dim3(grid), dim3(TAP_BLOCK), 0, s_, ROLL_HOT_ARGS(a), a); \
            else if (nw == 3) hipLaunchKernelGGL((k_rolling_window_wide<D_, 3>), dim3(grid), dim3(TAP_BLOCK), 0, s_, ROLL_HOT_ARGS(a), a); \
            else hipLaunchKernelGGL((k_rolling_window_wide<D_, 4>), dim3(grid), dim3(TAP_BLOCK), 0, s_, ROLL_HOT_ARGS(a), a); } while (0)
        if (D == 2) TAP_ROLL_WIDE(2); else TAP_ROLL_WIDE(3);
#undef TAP_ROLL_WIDE
        TAP_LAUNCH_CHECK(ctx, "k_rolling_window(wide)");
        return TAP_OK;
    }
    if (N > 64) {                                                     // one thread per instance
        const int g2 = (B + 63) / 64;
        hipStream_t s_ = (hipStream_t)stream;
#define TAP_ROLL_BIG(D_) do { \
            if (N <= 256) hipLaunchKernelGGL((k_rolling_window_big<D_, 4>), dim3(g2), dim3(64), 0, s_, a); \
            else if (N <= 1024) hipLaunchKernelGGL((k_rolling_window_big<D_, 16>), dim3(g2), dim3(64), 0, s_, a); \
            else hipLaunchKernelGGL((k_rolling_window_big<D_, 64>), dim3(g2), dim3(64), 0, s_, a); } while (0)
        if (D == 2) TAP_ROLL_BIG(2); else TAP_ROLL_BIG(3);
#undef TAP_ROLL_BIG
        TAP_LAUNCH_CHECK(ctx, "k_rolling_window_big");
        return TAP_OK;
    }
    const bool fast = roll_fast_ok(D, child);
    if (D == 2) {
        if (fast) hipLaunchKernelGGL((k_rolling_window<2, 10>), dim3(grid), dim3(TAP_BLOCK), 0, (hipStream_t)stream, ROLL_HOT_ARGS(a), a);
        else hipLaunchKernelGGL((k_rolling_window<2, 0>), dim3(grid), dim3(TAP_BLOCK), 0, (hipStream_t)stream, ROLL_HOT_ARGS(a), a);
    } else {
        if (fast) hipLaunchKernelGGL((k_rolling_window<3, 10>), dim3(grid), dim3(TAP_BLOCK), 0, (hipStream_t)stream, ROLL_HOT_ARGS(a), a);
        else hipLaunchKernelGGL((k_rolling_window<3, 0>), dim3(grid), dim3(TAP_BLOCK), 0, (hipStream_t)stream, ROLL_HOT_ARGS(a), a);
    }
    TAP_LAUNCH_CHECK(ctx, "k_rolling_window");
    return TAP_OK;
}

template <int D, int G> static int launch_rolling_step(tap_ctx *ctx, const RollStepArgs &a, hipStream_t st)
{
    constexpr int EPB = ROLL_EPB, ENV_WAVES = (EPB * G + 63) / 64, THREADS = 64 * (ENV_WAVES + EPB);
    const int grid = (a.r.B + EPB - 1) / EPB;
    if (grid == 0) return TAP_OK;
    const bool hard = a.s.d.flags & TAP_F_HARD;              // soft rewards: no hard-mode walk compiled in
    if (a.r.N > 64) {
        // (soft rewards too: the soft-only kernel's 64-register cap costs the two-word window code 20 spilled registers
        //  and 24.6 against 21.1 us per step at N = 100, B = 8192; this one is held to 72)
        hipLaunchKernelGGL((k_rolling_step<D, G, ROLL_CH_WIDE>), dim3(grid), dim3(THREADS), 0, st, ROLL_HOT_ARGS(a.r), a);
    } else if (roll_fast_ok(D, a.r.child)) {
        if (hard) hipLaunchKernelGGL((k_rolling_step<D, G, 10>), dim3(grid), dim3(THREADS), 0, st, ROLL_HOT_ARGS(a.r), a);
        else hipLaunchKernelGGL((k_rolling_step_soft<D, G, 10>), dim3(grid), dim3(THREADS), 0, st, ROLL_HOT_ARGS(a.r), a);
    } else {
        if (hard) hipLaunchKernelGGL((k_rolling_step<D, G, 0>), dim3(grid), dim3(THREADS), 0, st, ROLL_HOT_ARGS(a.r), a);
        else hipLaunchKernelGGL((k_rolling_step_soft<D, G, 0>), dim3(grid), dim3(THREADS), 0, st, ROLL_HOT_ARGS(a.r), a);
    }
    TAP_LAUNCH_CHECK(ctx, "k_rolling_step");
    return TAP_OK;
}

// by-products of a roller's step (tap_common.h: StepArgs aux fields) + the sticky error flag
struct RollAux {
    float *dec_static_out;
    int64_t *tour_out;
    int32_t *picked_out;
    const int32_t *nodes_cur;
    int tour_stride, tour_col;
};

static int rolling_step_impl(tap_ctx *ctx, const tap_env_desc *d, void *env_state, int N, int child,
                             const int32_t *blocks, const uint64_t *rel, uint64_t *state,
                             const int64_t *ptr, const float *static_cur, float *static_next,
                             float *dynamic_out, float *colsum_out, uint64_t *bits_out,
                             float *current_mask_out, int32_t *nodes_out, int32_t *err_out,
                             float *feature_out, void *stream, const RollAux *aux)
{
    int rc = tap_desc_validate(ctx, d);
    if (rc) return rc;
    if (d->B == 0) return TAP_OK; // an empty batch has no buffers to check
    rc = roll_check(ctx, d->B, d->D, N, child);
    if (rc) return rc;
    if (!env_state || !blocks || !rel || !state || !ptr || !static_cur || !static_next || static_cur == static_next)
        return tap_fail(ctx, TAP_E_INVALID, "bad rolling_step arguments (static_cur and static_next must differ)");
    if (!dynamic_out && !bits_out)
        return tap_fail(ctx, TAP_E_INVALID, "dynamic_out may only be left out when bits_out takes the window's bit shadow");
    if (d->strategy != TAP_LB_GREEDY || tap_is_big(d) || (N > 64 && roll_wide_nw(N, child) != 2)) {
        // no single kernel for these (MACS / legacy LB placements, thread-per-container shapes, instances above 128
        // blocks: their window wave keeps up to 16 mask words per lane, the fused kernels are held to 72 registers):
        // the same step as its two launches
        rc = tap_env_step_gather(ctx, d, env_state, static_cur, 1 + d->D, child * (d->D == 2 ? 2 : 6), ptr, nullptr,
                                 feature_out, stream);
        if (rc == TAP_OK && aux) {
            StepArgs s = {};
            s.d = *d; s.static_ = static_cur; s.static_rows = 1 + d->D; s.nR = child * (d->D == 2 ? 2 : 6); s.ptr = ptr;
            s.dec_static_out = aux->dec_static_out; s.tour_out = aux->tour_out; s.picked_out = aux->picked_out;
            s.nodes_cur = aux->nodes_cur; s.child = child; s.tour_stride = aux->tour_stride; s.tour_col = aux->tour_col;
            rc = tap_step_aux_launch(ctx, s, (hipStream_t)stream);
        }
        return rc ? rc : rolling_window_impl(ctx, d->B, d->D, N, child, blocks, rel, state, ptr, static_next, dynamic_out,
                                             colsum_out, bits_out, current_mask_out, nodes_out, err_out, stream, aux != nullptr);
    }
    RollStepArgs a = {};
    const int R = d->D == 2 ? 2 : 6;
    a.r.err_sticky = aux != nullptr;
    a.r.wt = tap_write_through((size_t)d->B * 3 * child * child * R * sizeof(float));
    a.r.B = d->B; a.r.D = d->D; a.r.N = N; a.r.child = child; a.r.blocks = blocks;
    a.r.rel = reinterpret_cast<unsigned long long *>(const_cast<uint64_t *>(rel));
    a.r.state = reinterpret_cast<unsigned long long *>(state);
    a.r.remove_ptr = ptr; a.r.static_out = static_next; a.r.dynamic_out = dynamic_out;
    a.r.colsum_out = colsum_out; a.r.cur_mask_out = current_mask_out; a.r.nodes_out = nodes_out; a.r.err_out = err_out;
    a.r.bits_out = reinterpret_cast<unsigned long long *>(bits_out);
    if (bits_out && (3 * child > 64 || (child * R) % 4 != 0 || child * R > 256))
        return tap_fail(ctx, TAP_E_UNSUPPORTED, "bits_out needs 3*child <= 64, (child*R) %% 4 == 0, child*R <= 256");
    a.s.d = *d;
    tap_env_layout(d, env_state, &a.s.v);
    a.s.static_ = static_cur; a.s.static_rows = 1 + d->D; a.s.nR = child * R; a.s.ptr = ptr;
    a.s.feature_out = feature_out; a.s.flen = tap_env_feature_len(d);
    a.s.lut = ctx ? ctx->stab_lut : nullptr;
    if (aux) {
        a.s.dec_static_out = aux->dec_static_out; a.s.tour_out = aux->tour_out; a.s.picked_out = aux->picked_out;
        a.s.nodes_cur = aux->nodes_cur; a.s.child = child; a.s.tour_stride = aux->tour_stride; a.s.tour_col = aux->tour_col;
    }
    const int Gs = tap_group_size(d);
    hipStream_t st = (hipStream_t)stream;
    if (d->D == 2) {
        if (Gs == 8) return launch_rolling_step<2, 8>(ctx, a, st);
        if (Gs == 16) return launch_rolling_step<2, 16>(ctx, a, st);
        if (Gs == 32) return launch_rolling_step<2, 32>(ctx, a, st);
        return launch_rolling_step<2, 64>(ctx, a, st);
    }
    if (Gs == 8) return launch_rolling_step<3, 8>(ctx, a, st);
    if (Gs == 16) return launch_rolling_step<3, 16>(ctx, a, st);
    if (Gs == 32) return launch_rolling_step<3, 32>(ctx, a, st);
    return launch_rolling_step<3, 64>(ctx, a, st);
}

extern "C" int tap_rolling_step(tap_ctx *ctx, const tap_env_desc *d, void *env_state, int N, int child,
                                const int32_t *blocks, const uint64_t *rel, uint64_t *state,
                                const int64_t *ptr, const float *static_cur, float *static_next,
                                float *dynamic_out, float *colsum_out, uint64_t *bits_out,
                                float *current_mask_out, int32_t *nodes_out, int32_t *err_out,
                                float *feature_out, void *stream)
{
    return rolling_step_impl(ctx, d, env_state, N, child, blocks, rel, state, ptr, static_cur, static_next, dynamic_out,
                             colsum_out, bits_out, current_mask_out, nodes_out, err_out, feature_out, stream, nullptr);
}

// ---- tap_roller: the step object of rolling.validate's loop (tapenv.h) --------------------------------------------
// Host-side only, like tap_stepper (transition.hip): remembers the caller's buffers, alternates the two phases of
// the window's static tensor and node list, and has the step's launch write decoder_static, the tour column and
// the picked block's global id.
struct tap_roller {
    tap_ctx *ctx;
    tap_env_desc d;
    void *env_state;
    int N, child;
    tap_roller_buffers b;
    const int32_t *blocks;
    const uint64_t *rel;
    uint64_t *state;
    int k;   // windows emitted so far minus one = index of the next step; -1 before begin
};

__global__ void __launch_bounds__(TAP_BLOCK) k_roll_zero_i32(int32_t *p, int n)
{
    const int i = blockIdx.x * TAP_BLOCK + threadIdx.x;
    if (i < n) p[i] = 0;
}

struct RollDeviceGuard {
    int prev, want;
    explicit RollDeviceGuard(int dev) : prev(-1), want(dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != want) (void)hipSetDevice(want);
    }
    ~RollDeviceGuard() { if (prev >= 0 && prev != want) (void)hipSetDevice(prev); }
};

extern "C" int tap_roller_create(tap_ctx *ctx, const tap_env_desc *d, void *env_state, int N, int child,
                                 const tap_roller_buffers *buf, tap_roller **out)
{
    if (!ctx || !d || !buf || !out) return TAP_E_INVALID;
    int rc = tap_desc_validate(ctx, d);
    if (rc) return rc;
    rc = roll_check(ctx, d->B, d->D, N, child);
    if (rc) return rc;
    if (d->B > 0 && (!env_state || !buf->static_[0] || !buf->static_[1] || buf->static_[0] == buf->static_[1] ||
                     (!buf->dynamic && !buf->bits) || !buf->nodes[0] || !buf->nodes[1] || buf->nodes[0] == buf->nodes[1]))
        return tap_fail(ctx, TAP_E_INVALID, "roller needs both phases of static / nodes (distinct) and dynamic (or, without the fp32 expansion, bits)");
    if ((buf->tour || buf->picked) && buf->tour_stride < N - child)
        return tap_fail(ctx, TAP_E_INVALID, "roller tour_stride must hold the N - child single-step windows");
    tap_roller *r = new (std::nothrow) tap_roller();
    if (!r) return tap_fail(ctx, TAP_E_INVALID, "out of host memory");
    r->ctx = ctx; r->d = *d; r->env_state = env_state; r->N = N; r->child = child; r->b = *buf;
    r->blocks = nullptr; r->rel = nullptr; r->state = nullptr; r->k = -1;
    *out = r;
    return TAP_OK;
}

extern "C" void tap_roller_destroy(tap_roller *r) { delete r; }

extern "C" int tap_roller_begin(tap_roller *r, const int32_t *blocks, const uint64_t *rel, uint64_t *state, void *stream)
{
    if (!r) return TAP_E_INVALID;
    if (r->d.B == 0) { r->k = 0; return TAP_OK; }
    if (!blocks || !rel || !state) return tap_fail(r->ctx, TAP_E_INVALID, "roller_begin needs blocks, rel and state");
    r->blocks = blocks; r->rel = rel; r->state = state; r->k = -1;
    RollDeviceGuard g(r->ctx->device);
    if (r->b.err) {
        hipLaunchKernelGGL(k_roll_zero_i32, dim3((r->d.B + TAP_BLOCK - 1) / TAP_BLOCK), dim3(TAP_BLOCK), 0, (hipStream_t)stream,
                           r->b.err, r->d.B);
        TAP_LAUNCH_CHECK(r->ctx, "k_roll_zero_i32");
    }
    const int rc = rolling_window_impl(r->ctx, r->d.B, r->d.D, r->N, r->child, blocks, rel, state, nullptr, r->b.static_[0],
                                       r->b.dynamic, r->b.colsum, reinterpret_cast<uint64_t *>(r->b.bits), r->b.current_mask,
                                       r->b.nodes[0], r->b.err, stream, 1);
    if (rc == TAP_OK) r->k = 0;
    return rc;
}

extern "C" int tap_roller_step(tap_roller *r, const int64_t *ptr, void *stream)
{
    if (!r) return TAP_E_INVALID;
    if (r->k < 0) return tap_fail(r->ctx, TAP_E_INVALID, "tap_roller_begin has not been called");
    if (r->k >= r->N - r->child) return tap_fail(r->ctx, TAP_E_STEPS, "all %d single-step windows are done", r->N - r->child);
    if (r->d.B == 0) { r->k += 1; return TAP_OK; }
    if (!ptr) return tap_fail(r->ctx, TAP_E_INVALID, "null ptr");
    const int k = r->k, cur = k & 1, nxt = cur ^ 1;
    const RollAux aux = {r->b.decoder_static, r->b.tour, r->b.picked, r->b.nodes[cur], r->b.tour_stride, k};
    RollDeviceGuard g(r->ctx->device);
    const int rc = rolling_step_impl(r->ctx, &r->d, r->env_state, r->N, r->child, r->blocks, r->rel, r->state, ptr,
                                     r->b.static_[cur], r->b.static_[nxt], r->b.dynamic, r->b.colsum,
                                     reinterpret_cast<uint64_t *>(r->b.bits), r->b.current_mask, r->b.nodes[nxt], r->b.err,
                                     r->b.feature, stream, &aux);
    if (rc == TAP_OK) r->k = k + 1;
    return rc;
}

extern "C" int tap_roller_steps_done(const tap_roller *r) { return r ? r->k : TAP_E_INVALID; }
