/*
 * tapenv.h -- C ABI of libtapenv.so: the MI355X (gfx950) batched Transport-and-Pack environment.
 *
 * This is the drop-in boundary for the reference's (Juzhan/TAP-Net) packing hot path.  The
 * reference has no FFI -- its seams are plain Python callables and one class -- so every entry
 * point below names the reference symbol (file:line) it replaces; INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - Plain C: pointers and sizes only, no torch / HIP types.  `stream` is a hipStream_t passed
 *     as void* (NULL = the null stream).  Every call is asynchronous on `stream` unless it says
 *     "synchronous".
 *   - All data pointers are DEVICE pointers; the caller owns every buffer (e.g. torch tensors).
 *     The library allocates nothing per call and keeps no reference to caller memory.
 *   - Return value: TAP_OK (0) or a negative TAP_E_* code; tap_last_error(ctx) gives a message.
 *   - A tap_ctx is bound to one device and is not thread-safe; distinct contexts are independent.
 *   - There is no CPU fallback: without a HIP device tap_ctx_create fails with TAP_E_NODEVICE.
 *
 * Per-env state is an opaque device blob of tap_env_state_bytes(desc) bytes holding only the
 * height-map, four counters, an error word and the placement history (layout: DESIGN.md).
 */
#ifndef TAPENV_H
#define TAPENV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TAP_ABI_VERSION 1

typedef struct tap_ctx tap_ctx;

enum {
    TAP_OK = 0,
    TAP_E_INVALID = -1,     /* bad argument */
    TAP_E_UNSUPPORTED = -2, /* valid in the reference, not implemented here (e.g. > 4096 cells) */
    TAP_E_HIP = -3,         /* HIP runtime error */
    TAP_E_OVERFLOW = -4,    /* a placement reached above H (reference: IndexError tools.py:2109 /
                               silent clipping tools.py:2169) */
    TAP_E_NODEVICE = -5,    /* no usable HIP device */
    TAP_E_STEPS = -6        /* add_new_block called more than blocks_num times (tools.py:3677) */
};

/* packing strategy, tools.py:3617-3620 / 3679-3690 */
enum { TAP_LB_GREEDY = 0, TAP_MACS = 1 /* 'MACS' and 'MUL' */,
       TAP_LB = 2 /* the legacy 'LB' (tools.py:1602-1955): voxel-level state, tap_env_step only */ };

/* flag word: the string tests the reference performs on reward_type */
enum {
    TAP_F_HARD = 1 << 0,     /* reward_type.endswith('hard')  tools.py:2113 */
    TAP_F_USE_P = 1 << 1,    /* 'P' in reward_type             tools.py:2135 */
    TAP_F_USE_S = 1 << 2,    /* 'S' in reward_type             tools.py:2138 */
    TAP_F_MCS_ZERO = 1 << 3, /* reward_type.startswith('mcs')  tools.py:2709 */
    TAP_F_MCS_TIE = 1 << 4   /* 'mcs' in reward_type           tools.py:2718 */
};

/* Container.calc_ratio formula, tools.py:3907-3966 */
enum {
    TAP_R_C = 0, TAP_R_CxS, TAP_R_CP, TAP_R_CPxS, TAP_R_CPS, TAP_R_2CPS, TAP_R_CxPxS,
    TAP_R_CP_HALF /* 'C+P-lb-soft' -> (C+P)/2, every other mode divides by 3 */
};

/* heightmap_type, tools.py:3716-3744 */
enum { TAP_FEAT_FULL = 0, TAP_FEAT_ZERO = 1, TAP_FEAT_DIFF = 2 };

/* element type of the `blocks` argument */
enum { TAP_DT_F32 = 0, TAP_DT_I32 = 1 };

typedef struct tap_env_desc {
    int32_t B;          /* number of containers stepped in lock-step */
    int32_t D;          /* 2 | 3 */
    int32_t W, L, H;    /* container_size; L = 1 when D == 2 */
    int32_t n_max;      /* blocks_num */
    int32_t strategy;   /* TAP_LB_GREEDY | TAP_MACS | TAP_LB */
    int32_t flags;      /* TAP_F_* */
    int32_t ratio_mode; /* TAP_R_* */
    int32_t feature;    /* TAP_FEAT_* */
} tap_env_desc;

/* ---- library / context ---------------------------------------------------------------- */

int tap_abi_version(void);
const char *tap_status_string(int status);
/* synchronous.  device = HIP ordinal. */
int tap_ctx_create(int device, tap_ctx **out);
void tap_ctx_destroy(tap_ctx *ctx);
const char *tap_last_error(const tap_ctx *ctx);

/* ---- tools.Container, batched --------------------------------------------------------- */

/* Fill `d` from the arguments of tools.Container.__init__ (tools.py:3611-3661), performing its
 * string tests once on the host: container_size = D ints, reward_type e.g. "C+P+S-lb-soft",
 * heightmap_type "full"|"zero"|"diff", packing_strategy "LB_GREEDY"|"MACS"|"MUL"|"LB" (the reward
 * string overrides it exactly as tools.py:3617-3620 does).  Host only, no device work. */
int tap_env_desc_init(tap_env_desc *d, int B, int D, const int32_t *container_size, int blocks_num,
                      const char *reward_type, const char *heightmap_type,
                      const char *packing_strategy);

/* bytes of device memory the caller must provide for the state blob (256-byte aligned) */
size_t tap_env_state_bytes(const tap_env_desc *d);
/* floats per env returned by step/feature: W*L (full/zero), W-1 (2D diff), 2*W*L (3D diff) */
int tap_env_feature_len(const tap_env_desc *d);

/* tools.Container.__init__ state / clear_container (tools.py:3629-3655, 3858-3885).  The blob is cleared by a
 * kernel, not hipMemsetAsync: the call may be captured into a hipGraph (a captured memset becomes a graph memset
 * node, which was observed to run out of order with the neighbouring kernel nodes on replay).  `state` must be
 * 16-byte aligned (TAP_E_INVALID otherwise). */
int tap_env_reset(tap_ctx *ctx, const tap_env_desc *d, void *state, void *stream);

/* tools.Container.add_new_block for all B envs (tools.py:3663-3744 -> calc_one_position_lb_greedy
 * 2027-2351, is_stable_2d 839-868, is_stable 710-765; with strategy TAP_MACS ->
 * calc_one_position_mcs_2d 2456-2749 / calc_one_position_mcs_3d 2751-3165 (2D: W <= 4096, H <= 4096 -- lane-per-column
 * kernels up to 16 columns, one wavefront per container above; 3D: W, L <= 64, H <= 4096 -- lane-per-cell kernel up to
 * 64 cells with sides <= 8, one wavefront per container above; block sides <= container sides, else error bit 4);
 * on the wave-per-container paths (LB_GREEDY and MACS 3D above 64 cells) tools.is_stable is evaluated for block
 * footprints up to 16 x 16 (csrc/tap_stable_wide.h beyond the 8 x 8 support masks), larger ones raise error bit 4;
 * model.py:451-465 is the loop it replaces).
 *   blocks      (B, D) TAP_DT_F32 | TAP_DT_I32, one block per env; f32 is truncated like
 *               block.astype(int) (tools.py:3689)
 *   active      (B,) uint8 or NULL: envs with 0 are not stepped and only report their feature
 *               (the 'mul' input types' idle container, model.py:419-427)
 *   feature_out (B, feature_len) f32 or NULL: what add_new_block returns, already in the layout
 *               model.py:456-465 builds ((B,W-1,1) / (B,2,W,L) for 'diff') */
int tap_env_step(tap_ctx *ctx, const tap_env_desc *d, void *state, const void *blocks,
                 int blocks_dtype, const uint8_t *active, float *feature_out, void *stream);

/* Same, with the gather of model.py:404-412 fused in: block[k] = static_[b, 1+k, ptr[b]].
 * static_ (B, static_rows, nR) f32, ptr (B,) int64. */
int tap_env_step_gather(tap_ctx *ctx, const tap_env_desc *d, void *state, const float *static_,
                        int static_rows, int nR, const int64_t *ptr, const uint8_t *active,
                        float *feature_out, void *stream);

/* tools.Container.get_heightmap (tools.py:3824-3856): the feature of the current state */
int tap_env_feature(tap_ctx *ctx, const tap_env_desc *d, const void *state, float *feature_out,
                    void *stream);

/* tools.Container.calc_ratio / calc_CPS (tools.py:3887-3966; model.py:499-510 stores it as fp32).
 * ratio_out (B,) f32, ratio64_out (B,) f64, cps_out (B,3) f64 -- each nullable. */
int tap_env_ratio(tap_ctx *ctx, const tap_env_desc *d, const void *state, float *ratio_out,
                  double *ratio64_out, double *cps_out, void *stream);

/* Container attributes in the reference's layouts (all nullable):
 * heightmap (B, W*L) i32; positions (B, n_max, D) i32; stable (B, n_max) u8;
 * counters (B, 4) i32 = valid_size, empty_size, sum(stable), current_blocks_num. */
int tap_env_export(tap_ctx *ctx, const tap_env_desc *d, const void *state, int32_t *heightmap_out,
                   int32_t *positions_out, uint8_t *stable_out, int32_t *counters_out,
                   void *stream);

/* tools.is_stable (tools.py:710-765) evaluated mask by mask, for tests of the predicate itself: the
 * placement kernels call the same two device functions.  masks (n,) uint64, bit (i*by + j) = footprint
 * cell (i, j) of a bx x by block rests on a voxel (tools.py:722-728); the block is off the floor.
 * use_lut != 0: through the per-context table the 3D kernels use for footprints <= 4x4 (larger ones
 * fall through to the direct form, as in the kernels); 0: the direct hull-free form only.
 * stable_out (n,) uint8.  bx, by <= 8. */
int tap_stable3d_eval(tap_ctx *ctx, int bx, int by, const unsigned long long *masks, int n,
                      int use_lut, uint8_t *stable_out, void *stream);

/* The containers' sticky error words, asynchronously: err_out (B,) int32 with bit 1 = a placement reached above H
 * (the reference raises IndexError, tools.py:2109), 2 = add_new_block called more than blocks_num times, 4 = bad
 * block / column index, 8 / 16 = MACS list guards.  For callers that report per container instead of raising. */
int tap_env_errors(tap_ctx *ctx, const tap_env_desc *d, const void *state, int32_t *err_out, void *stream);

/* Synchronous.  Returns TAP_OK, or TAP_E_OVERFLOW / TAP_E_STEPS if any env has raised its sticky
 * error word; *n_bad_out (host, nullable) = number of such envs. */
int tap_env_check(tap_ctx *ctx, const tap_env_desc *d, const void *state, int32_t *n_bad_out,
                  void *stream);

/* pack.reward (pack.py:378-473) == tools.calc_positions_lb_greedy (tools.py:2393-2449) per env:
 * the whole episode in one launch.  static_ (B, static_rows, nR) f32, tour (B, n) int64,
 * reward_out (B,) f32 = -(C+P+S) un-normalised; positions_out (B, n, D) i32 and stable_out
 * (B, n) u8 nullable.  d->B is ignored (B given here); no state blob is needed. */
int tap_episode_reward(tap_ctx *ctx, const tap_env_desc *d, int B, int n, const float *static_,
                       int static_rows, int nR, const int64_t *tour, float *reward_out,
                       int32_t *positions_out, uint8_t *stable_out, void *stream);

/* tools.calc_positions_lb_greedy (tools.py:2393-2449) / tools.calc_positions_mcs (tools.py:3213-3315) per env,
 * as pack.render calls them for its metric files (pack.py:743-792; caller trainer.py:132, 493): the whole
 * episode in one launch with every strategy tap_env_step has except the legacy 'LB' (render never uses it,
 * pack.py:741), containers up to 64 cells.  static_ / tour as tap_episode_reward.
 *   target_sel   -1: every tour entry.  0 | 1: the two-container input types ('mul', 'mul-with', pack.py:755-790):
 *                only the entries whose target id -- the LAST row of static_ -- equals it, in tour order.
 *   ratio64_out  (B,) f64: the function's `ratio` (tools.py:2442-2445 C+P+S; tools.py:3279-3308 by reward type,
 *                d->ratio_mode without Container.calc_ratio's division); 0 for an empty list (pack.py:760-769);
 *                NaN when the container raised an error bit
 *   scores_out   (B, 5) int64: valid_size, box_size, empty_size, stable_num, max(heightmap) (tools.py:2447, 3311);
 *                zeros for an empty list
 *   err_out      (B,) int32: the sticky error bits tap_env_check reports (1 overflow, 4 bad block / index, 8 / 16 MACS)
 * every output is nullable; positions_out (B, n, D) / stable_out (B, n) are indexed by tour position (entries of
 * the other container stay 0). */
int tap_episode_scores(tap_ctx *ctx, const tap_env_desc *d, int B, int n, const float *static_,
                       int static_rows, int nR, const int64_t *tour, int target_sel, double *ratio64_out,
                       int64_t *scores_out, int32_t *positions_out, uint8_t *stable_out, int32_t *err_out,
                       void *stream);

/* ---- instance generation (generate.py:773-971) ---------------------------------------- */

/* tools.calc_positions_lb_greedy (tools.py:2393-2449) for B explicit block lists: blocks (B, n, D)
 * int32 in placement order, one launch.  This is what generate.generate_blocks calls with
 * 'C+P+S-lb-hard' on the initial container (generate.py:908); an instance is accepted when every
 * stable_out flag is 1 (generate.py:909-910).  With strategy TAP_MACS: tools.calc_positions_mcs
 * (tools.py:3213-3315).  LB_GREEDY containers of any size (generate.py:908 accepts any --initial_container_width):
 * lane-per-cell groups up to 64 cells, above that one wavefront per container with the height-map in LDS across the n
 * placements (big.hip: k_big_wave_episode; tap_episode_reward / tap_episode_scores take the same path); MACS / MUL
 * above 64 cells: TAP_E_UNSUPPORTED, step them with tap_env_step_gather.  reward_out (B,) f32 = -(C+P+S), positions_out
 * (B, n, D) i32, stable_out (B, n) u8, score64_out (B,) f64 = C+P+S -- each nullable.
 * A block with a side < 1 is not part of its list (lists of different length in one batch: the
 * two-container reward of pack.py:451-466 packs the blocks of each target id separately); S is
 * taken over the blocks that are, and an empty list scores 0 (pack.py:459-460). */
int tap_pack_blocks(tap_ctx *ctx, const tap_env_desc *d, int B, int n, const int32_t *blocks,
                    float *reward_out, int32_t *positions_out, uint8_t *stable_out,
                    double *score64_out, void *stream);

/* generate.calc_dependent (generate.py:575-771) + the rotation bookkeeping of
 * generate.generate_blocks (generate.py:935-971) + pack.PACKDataset's layout (pack.py:101-195,
 * input_type 'bot', allow_rot=True): from blocks and positions (B, n, D) i32 of fully packed
 * initial containers to static_out (B, 1+D, n*R) and dynamic_out (B, 3n, n*R) f32.
 * container_size = D host ints (the INITIAL container); arm_size as generate.py:623-641 (2D).
 * n <= 64. */
int tap_precedence(tap_ctx *ctx, int B, int D, int n, const int32_t *container_size, int arm_size,
                   const int32_t *blocks, const int32_t *positions, float *static_out,
                   float *dynamic_out, void *stream);

/* ---- perfect-packing ("PPSG") instances (generate.py:17-301) -------------------------------- */
/* Random draws are made the way numpy's RandomState makes them (random_sample, masked-rejection randint,
 * choice(p) by searchsorted on the fp64 cumulative sum) on counter-based word streams: word i of the stream
 * with key k = high 32 bits of splitmix64-finalise(k + i * 0x9E3779B97F4A7C15), k = key(seed, a, b, c) as
 * csrc/ppsg.hip / oracle/tap_oracle.c define it.  ids (B,) int64 = global instance ids (nullable:
 * instance0 + b), so any sharding generates the same instances. */

/* generate.BPP_Generator_3D (generate.py:232-301) inside generate_blocks_with_GT's acceptance loop
 * (generate.py:66-73): for every instance S slabs of ns <= 16 blocks, slab s = n - 1 guillotine cuts of a
 * W x W x heights[b, s] box, re-drawn (attempt a = 0, 1, ...; stream key(seed, id*S + s, gen, a)) until every
 * side is in [min_size, max_size); the slabs are stacked along z -- S = 1 is the reference's generator,
 * S > 1 builds perfect packings of S*ns blocks, which the reference's single rejection loop cannot reach
 * (acceptance < 2e-8 at 50 blocks).  gt_blocks_out / gt_positions_out (B, S*ns, 3) i32; attempts_out (B, S)
 * i32 nullable (-1 = max_attempts reached, that slab is then invalid). */
int tap_ppsg_gt(tap_ctx *ctx, int B, int S, int ns, int W, const int32_t *heights, int min_size,
                int max_size, uint64_t seed, const int64_t *ids, int64_t instance0, int gen,
                int64_t max_attempts, int32_t *gt_blocks_out, int32_t *gt_positions_out,
                int32_t *attempts_out, void *stream);

/* generate.py:86-105: a random order in which the perfect packing can be taken apart from the top and a
 * random rotation per block (stream key(seed, id, gen, 1000 + trial)) -> blocks_out (B, n, 3) i32 in layout
 * order, to be packed into the initial container with tap_pack_blocks ('C+P+S-lb-hard').  n <= 64. */
int tap_ppsg_order(tap_ctx *ctx, int B, int n, const int32_t *gt_blocks, const int32_t *gt_positions,
                   uint64_t seed, const int64_t *ids, int64_t instance0, int gen, int trial,
                   int32_t *blocks_out, void *stream);

/* The 2D forms (generate_blocks_with_GT with block_dim 2, generate.py:72-73): tap_ppsg_gt2d = generate.
 * BPP_Generator_2D_easy (generate.py:392-484) for B instances -- n - 1 guillotine cuts of a W x heights[b] box
 * (volume-weighted choice among the blocks with a side >= max_size, else among the splittable ones; the axis
 * rule of :446-454; uniform split positions for short sides, Gaussian ones from `gauss` for sides
 * >= 2*max_size - 1), re-drawn (stream key(seed, id, gen, attempt)) until every side is in [min_size, max_size).
 * gauss (gauss_rows, gauss_stride) f64, device: row L = the normalised cumulative sum np.random.choice makes
 * of the reference's Gaussian weights over the L - 2*min_size split positions of a side of length L (the
 * caller builds it with numpy, tap-net_amd/generate.py gauss_split_table; rows below 2*max_size - 1 unused).
 * gt_blocks_out / gt_positions_out (B, n, 2) i32; attempts_out (B,) i32 nullable, -1 = cap reached.
 * tap_ppsg_order2d = tap_ppsg_order on (B, n, 2) arrays with calc_dependent's 2D movement rule
 * (generate.py:575-647) and the two 2D rotations.  The layout test is tap_ppsg_check on the relations of the
 * 2D layout (forward / backward masks are empty there). */
int tap_ppsg_gt2d(tap_ctx *ctx, int B, int n, int W, const int32_t *heights, int min_size, int max_size,
                  const double *gauss, int gauss_stride, int gauss_rows, uint64_t seed, const int64_t *ids,
                  int64_t instance0, int gen, int64_t max_attempts, int32_t *gt_blocks_out,
                  int32_t *gt_positions_out, int32_t *attempts_out, void *stream);
int tap_ppsg_order2d(tap_ctx *ctx, int B, int n, const int32_t *gt_blocks, const int32_t *gt_positions,
                     uint64_t seed, const int64_t *ids, int64_t instance0, int gen, int trial,
                     int32_t *blocks_out, void *stream);

/* generate.py:110-156: accept a packed layout iff every block is stable (stable (B, n) u8 of
 * tap_pack_blocks) and the blocks can be taken out again last-packed-first (rel (B, 5, n) u64 of
 * tap_rolling_init: nothing on top, one free side per horizontal axis; input_simple != 0 ignores the
 * sides).  ok_out (B,) u8. */
int tap_ppsg_check(tap_ctx *ctx, int B, int n, int input_simple, const uint64_t *rel, const uint8_t *stable,
                   uint8_t *ok_out, void *stream);

/* ---- rolling precedence windows (generate.py:1589-1839, rolling.py:589-637) ------------- */

/* generate.InitialContainer.__init__: the five dependency graphs of B fully packed initial
 * containers with N <= 4096 blocks each, as column masks of NW = ceil(N/64) uint64 words with bit a of a
 * node j's mask = "block a blocks block j": rel_out holds 5*N*NW words per instance -- first the N movement
 * masks, then one record of four masks (left, right, forward, backward) per node, so that a step reads the
 * window nodes' side masks as one piece of a cache line each; state_out holds 2*NW words (entered, window).
 * Up to 64 blocks an instance is handled by one wavefront (lane = node), up to 256 blocks with windows of at
 * most 64 / NW nodes still by one wavefront (lane = NW nodes, NW-word masks; tap_rolling_step is one launch up to
 * 128 blocks and two above); wider windows and instances of 257 .. 4096 blocks by one thread per instance
 * (rolling.py:831 leaves --total_blocks_num free).  state_out is cleared. */
int tap_rolling_init(tap_ctx *ctx, int B, int D, int N, const int32_t *container_size, int arm_size,
                     const int32_t *blocks, const int32_t *positions, uint64_t *rel_out,
                     uint64_t *state_out, void *stream);

/* One step of rolling.validate's loop: InitialContainer.remove_block for the column picked in the
 * previous window (remove_ptr (B,) int64, NULL on the first call), then convert_to_input():
 * top the window up to `child` nodes and emit static_out (B, 1+D, child*R), dynamic_out
 * (B, 3*child, child*R) f32.  Optional by-products: colsum_out (B, 3, child*R) (the column sums
 * update_mask needs), bits_out (B, child*R) uint64 (the bit shadow of dynamic_out for
 * tap_transition_bits / tap_mask_step_bits; needs 3*child <= 64, child*R % 4 == 0), current_mask_out (B, child*R) (model.py:297-307), nodes_out (B, child) i32
 * (sorted global block ids = static's columns), err_out (B,) i32 (1 = window could not be filled).
 * dynamic_out may be NULL when bits_out is given: the window's precedence tensor then only exists as its bit shadow
 * (no fp32 expansion -- 7 200 of the 9 937 bytes a c5 step moves; for policies that read the shadow, as
 * tap_stepper_buffers.dyn = NULL for the decoding step).  tap_rolling_step and tap_roller_buffers.dynamic alike. */
int tap_rolling_window(tap_ctx *ctx, int B, int D, int N, int child, const int32_t *blocks,
                       const uint64_t *rel, uint64_t *state, const int64_t *remove_ptr,
                       float *static_out, float *dynamic_out, float *colsum_out, uint64_t *bits_out,
                       float *current_mask_out, int32_t *nodes_out, int32_t *err_out, void *stream);

/* One decoding step of rolling.validate in one launch: add_new_block for the column `ptr` picked
 * in the CURRENT window (block gathered from static_cur, the tensor tap_rolling_window / _step
 * wrote last) fused with remove_block + convert_to_input() for the NEXT window (written to
 * static_next, which must be a different buffer).  feature_out as in tap_env_step.  One kernel for LB_GREEDY on
 * containers of at most 64 cells and instances of at most 128 blocks (windows of at most 32 nodes above 64 blocks);
 * otherwise the two launches behind this entry. */
int tap_rolling_step(tap_ctx *ctx, const tap_env_desc *d, void *env_state, int N, int child,
                     const int32_t *blocks, const uint64_t *rel, uint64_t *state, const int64_t *ptr,
                     const float *static_cur, float *static_next, float *dynamic_out,
                     float *colsum_out, uint64_t *bits_out, float *current_mask_out, int32_t *nodes_out,
                     int32_t *err_out, float *feature_out, void *stream);

/* ---- precedence tensors (pack.py:276-376, model.py:297-307) --------------------------- */
/* dynamic is (B, rows, nR) f32 with rows = 3n ('bot', 'mul') or n ('simple', 'rot'); nR = n*R.
 * colsum is a (B, 3, nR) f32 shadow of the three per-section column sums of a dynamic tensor
 * (move / small / large, pack.py:324-326); sections beyond `rows` sum to 0. */

/* colsum_out = column sums of dynamic */
int tap_dyn_colsum(tap_ctx *ctx, int B, int n, int nR, int rows, const float *dynamic,
                   float *colsum_out, void *stream);

/* pack.update_dynamic (pack.py:333-376): dyn_out = dyn_in with rows real + n*i (i < update_rows)
 * zeroed, real = (long)static_[b, 0, ptr[b]].  Out of place.  If colsum_in/out are given the
 * shadow is updated incrementally (no reduction). */
int tap_update_dynamic(tap_ctx *ctx, int B, int n, int nR, int rows, int update_rows,
                       const float *dyn_in, const float *static_, int static_rows,
                       const int64_t *ptr, float *dyn_out, const float *colsum_in,
                       float *colsum_out, void *stream);

/* pack.update_mask (pack.py:276-331) from the column sums of the (already updated) dynamic.
 * ptr == NULL gives the initial mask of model.py:297-307 (mask_in may then be NULL = ones).
 * current_out = new_mask.float(), mask_out = chosen_mask. */
int tap_update_mask(tap_ctx *ctx, int B, int n, int R, const float *mask_in, const float *colsum,
                    const int64_t *ptr, float *current_out, float *mask_out, void *stream);

/* update_dynamic + update_mask in one launch (what model.py:376-386 does per step). */
int tap_mask_step(tap_ctx *ctx, int B, int n, int R, int rows, int update_rows,
                  const float *dyn_in, const float *static_, int static_rows, const int64_t *ptr,
                  const float *mask_in, const float *colsum_in, float *dyn_out, float *colsum_out,
                  float *current_out, float *mask_out, void *stream);

/* ---- the same two seams on a bit shadow of `dynamic` ------------------------------------ */
/* `dynamic` only ever holds 0 and 1 (the precedence matrices PACKDataset builds, pack.py:101-195;
 * update_dynamic only writes zeros, pack.py:370-374).  Carried as a (B, nR) uint64 shadow -- word j
 * of env b = column j, bit r = dynamic[b, r, j] != 0; for 65 <= rows <= 128 (windows of 22 .. 42 nodes) the
 * shadow is (B, 2, nR): plane 0 = rows 0..63, plane 1 = rows 64..127 -- a step no longer re-reads the
 * fp32 tensor: clearing the chosen rows is an AND, the column sums of pack.py:323-326 are popcounts,
 * and the fp32 tensor the network consumes next (model.py:378) is expanded from the bits, so the
 * step WRITES rows*nR*4 bytes per env and reads nR*8.  Outputs are bit-identical to tap_mask_step /
 * tap_transition for 0/1 input. */

/* uint64 words PER ENV of the shadow of a (rows, nR) window: nR up to 64 rows, 2 * nR for 65 .. 128 rows, 0 when the
 * shape has no shadow.  Every bits_in / bits_out buffer below holds B * tap_bits_words(rows, nR) words; the library
 * cannot see the allocation, so size it with this. */
int tap_bits_words(int rows, int nR);

/* bits_out (B, tap_bits_words(rows, nR)) = shadow of dynamic; *nonbinary_out (device int32, nullable, caller zeroes
 * it) is incremented by the number of elements that are neither 0 nor 1 -- the shadow is only valid at 0.
 * bits_out NULL: count only (then rows may exceed 128). */
int tap_dyn_bits(tap_ctx *ctx, int B, int nR, int rows, const float *dynamic,
                 unsigned long long *bits_out, int32_t *nonbinary_out, void *stream);

/* tap_mask_step on the shadow (bits_in / bits_out: B * tap_bits_words(rows, nR) words each).  Every output is nullable (at least one must be given) and mask_in
 * NULL means ones, so the same entry serves pack.update_dynamic alone (no mask outputs) and
 * pack.update_mask alone (update_rows = 0, masks only); ptr NULL (with update_rows = 0; static_ is then
 * unused) gives the initial mask of model.py:297-307.  A ptr outside [0, nR) -- the reference's gather
 * raises -- clears no row and removes no column, here and in tap_update_dynamic / tap_mask_step /
 * tap_transition* (whose placement half also raises error bit 4 for it).  Requires nR % 4 == 0, nR <= 256,
 * rows <= 128, 16-byte aligned buffers, bits_in != bits_out. */
int tap_mask_step_bits(tap_ctx *ctx, int B, int n, int R, int rows, int update_rows,
                       const unsigned long long *bits_in, const float *static_, int static_rows,
                       const int64_t *ptr, const float *mask_in, unsigned long long *bits_out,
                       float *dyn_out, float *current_out, float *mask_out, void *stream);

/* The FIRST step of an episode on a fresh fp32 `dynamic` (what PACKDataset / a DataLoader hands over,
 * pack.py:195): tap_mask_step_bits with the shadow built from dyn_in inside the launch (one read of the
 * tensor), so an episode needs no separate tap_dyn_bits pass.  bits_out receives the shadow of dyn_out;
 * *nonbinary_out (device int32, nullable, caller zeroes it) counts the elements of dyn_in that are neither
 * 0 nor 1 -- every output of the launch is only valid when it stays 0.  ptr NULL (update_rows = 0):
 * shadow + initial mask only. */
int tap_mask_step_first(tap_ctx *ctx, int B, int n, int R, int rows, int update_rows, const float *dyn_in,
                        const float *static_, int static_rows, const int64_t *ptr, const float *mask_in,
                        unsigned long long *bits_out, float *dyn_out, float *current_out, float *mask_out,
                        int32_t *nonbinary_out, void *stream);

/* ---- one whole lock-step in one launch --------------------------------------------------- */

enum {
    TAP_T_FRESH = 1, /* treat the state blob as freshly reset (Container.__init__, model.py:294) */
    TAP_T_RATIO = 2  /* also emit Container.calc_ratio after the step (model.py:499-510) */
};

/* Everything DRL.forward does around its policy network for one decoding step (model.py:376-465):
 * tap_mask_step + tap_env_step_gather fused, so the placement's latency hides under the HBM-bound
 * precedence update and a kernel boundary disappears.  n = blocks in the precedence window
 * (nR = n*R columns); d->n_max may be larger (rolling windows over one long-lived container).
 * One kernel for LB_GREEDY (2D/3D, up to 64 cells) and MACS/MUL (2D up to 16 columns; 3D up to 8 x 8), as long as a workgroup's candidate lists fit the device's LDS (160 KiB per workgroup on gfx950); every other
 * shape and strategy tap_env_step takes (legacy 'LB', LB_GREEDY and MACS 3D above 64 cells, MACS 2D above 16 columns, a MACS
 * container whose candidate lists do not fit a fused workgroup's LDS) runs the same step as its two launches behind
 * this entry -- except on the bit shadow (tap_transition_bits / _first, the stepper), where the wave-per-container
 * shapes are one launch too since round 5: a container's wavefront runs its own precedence slab, then its placement
 * (big.hip, macs_big.hip, macs3_big.hip; for MACS a fresh container and calc_ratio stay launches of their own at the
 * two ends of an episode).  feature_out nullable; ratio_out (B,) f32 required with
 * TAP_T_RATIO. */
int tap_transition(tap_ctx *ctx, const tap_env_desc *d, void *state, int n, int R, int rows,
                   int update_rows, const float *dyn_in, const float *static_, int static_rows,
                   const int64_t *ptr, const float *mask_in, const float *colsum_in,
                   float *dyn_out, float *colsum_out, float *current_out, float *mask_out,
                   float *feature_out, float *ratio_out, int flags, void *stream);

/* 1 when the tap_transition* entry points run a step of this shape as ONE kernel, 2 when they run it as the
 * precedence update followed by the placement (see above; for MACS the answer depends on the device's LDS).
 * on_bits != 0: tap_transition_bits / _first (the single kernels carry the one-word shadow: rows <= 64). */
int tap_transition_launches(const tap_ctx *ctx, const tap_env_desc *d, int n, int R, int rows, int on_bits);

/* tap_transition with the precedence update done on the bit shadow (tap_mask_step_bits).  The single kernels
 * carry the one-word shadow (rows <= 64); with the two-word shadow the step runs as its two launches. */
int tap_transition_bits(tap_ctx *ctx, const tap_env_desc *d, void *state, int n, int R, int rows,
                        int update_rows, const unsigned long long *bits_in, const float *static_,
                        int static_rows, const int64_t *ptr, const float *mask_in,
                        unsigned long long *bits_out, float *dyn_out, float *current_out,
                        float *mask_out, float *feature_out, float *ratio_out, int flags,
                        void *stream);

/* tap_transition_bits for the first step of an episode: the shadow is built from the fp32 tensor dyn_in
 * inside the launch (see tap_mask_step_first). */
int tap_transition_first(tap_ctx *ctx, const tap_env_desc *d, void *state, int n, int R, int rows,
                         int update_rows, const float *dyn_in, const float *static_, int static_rows,
                         const int64_t *ptr, const float *mask_in, unsigned long long *bits_out,
                         float *dyn_out, float *current_out, float *mask_out, float *feature_out,
                         float *ratio_out, int32_t *nonbinary_out, int flags, void *stream);

/* ---- the step object of a decoding loop ----------------------------------------------------- */
/* DRL.forward's loop (model.py:342-496) runs its policy network between the environment steps, so the loop is
 * issued from the host, step by step, and what the host pays per step counts.  A tap_stepper is
 * tap_transition_first / tap_transition_bits with everything that does not change during an episode resolved
 * once: the caller allocates TWO phases of the step's outputs (the step of index k writes phase k & 1 and reads
 * phase (k & 1) ^ 1), the stepper remembers the pointers, alternates the phases, starts from a fresh container at
 * step 0 and emits calc_ratio at step `steps` - 1 -- one call with two arguments per step, no allocation, no
 * per-step argument marshalling.  The buffers stay the caller's (the library keeps no memory, only the addresses;
 * they must outlive the stepper).  Host-side object, not thread-safe; launches are asynchronous on `stream`. */
typedef struct tap_stepper tap_stepper;

typedef struct tap_stepper_buffers {
    unsigned long long *bits[2]; /* (B, nR) uint64 -- (B, 2, nR) above 64 rows: the bit shadow of dyn[w] */
    float *dyn[2];               /* (B, rows, nR): update_dynamic's result (pack.py:333-376) as the fp32 tensor model.py:378
                                    feeds the encoder.  Both NULL: the step keeps `dynamic` as bits[w] only and skips the
                                    expansion (78 % of a c2 step's bytes) -- for callers that consume the shadow; masks,
                                    placements, features and ratio are the same either way.
                                    dyn[0] == dyn[1] (windows with a bit shadow): ONE tensor, updated IN PLACE -- step 0 writes
                                    all of it, every later step only the rows it clears (pack.py:370-374: the result differs
                                    from the input in rows real + n*i alone; the reference's clone, pack.py:368, is there for
                                    autograd).  For loops under no_grad (validation, serving): the previous steps' tensors
                                    are gone.  The lane-per-cell fused steps (containers of at most 64 cells, MACS 2D up
                                    to 16 columns) write 3 rows per step instead of 3n; every other step form writes the whole
                                    tensor into the one buffer -- the same values either way */
    float *current[2];           /* (B, nR): update_mask's new_mask.float() (pack.py:329-331) */
    float *mask[2];              /* (B, nR): update_mask's chosen_mask */
    float *feature;              /* (B, feature_len) nullable: add_new_block's return, layout of model.py:456-465 */
    float *decoder_static;       /* (B, D) nullable: static[:, 1:1+D, ptr], the gather of model.py:404-406 */
    float *ratio;                /* (B,): calc_ratio, written by the last step (model.py:499-510) */
    int64_t *tour;               /* (B, tour_stride) nullable: column tour_col0 + k = the ptr of step k (model.py:495, 512) */
    int32_t *nonbinary;          /* device int32, nullable, caller zeroes it: see tap_mask_step_first */
    int32_t tour_stride;         /* 0 = steps */
    int32_t tour_col0;           /* first tour column this stepper writes (a rolling episode's last window: N - child) */
    float *colsum[2];            /* (B, 3, nR), only for windows WITHOUT a bit shadow (nR % 4 != 0, nR > 256 or rows > 128;
                                    NULL otherwise): the column-sum shadow of dyn[w]; the step is then tap_transition's
                                    fp32-copy form, dyn[] is required and bits[] is unused */
} tap_stepper_buffers;

/* d / state: the containers (tap_env_desc_init, a blob of tap_env_state_bytes); n, R, rows, update_rows,
 * static_rows as in tap_transition_bits; steps = decoding steps per episode (model.py:342: blocks_num).
 * Windows without a bit shadow (nR % 4 != 0, nR > 256, rows > 128) run tap_transition's fp32-copy form behind the same
 * three calls (buf->colsum; tap_stepper_begin then always makes its two small launches; tap_stepper_begin_shadow is
 * TAP_E_UNSUPPORTED there). */
int tap_stepper_create(tap_ctx *ctx, const tap_env_desc *d, void *state, int n, int R, int rows,
                       int update_rows, int static_rows, int steps, const tap_stepper_buffers *buf,
                       tap_stepper **out);
void tap_stepper_destroy(tap_stepper *s);

enum {
    TAP_SB_INITIAL_MASK = 1, /* tap_stepper_begin: also produce the masks DRL.forward starts from */
    TAP_SB_CONTINUE = 2      /* step 0 goes on with the containers as they are (rolling.py:428: one long-lived
                                container across the windows); default: step 0 starts from fresh containers */
};

/* Bind the next instance batch: static_ (B, static_rows, nR) and the fresh fp32 dynamic (B, rows, nR) a
 * DataLoader hands over (pack.py:195); both are only read and must stay valid for the episode.
 * TAP_SB_INITIAL_MASK: one launch builds the shadow and the masks DRL.forward starts from (model.py:297-307)
 * into phase 1 -- current[1] is what the policy sees before step 0.  Without it: no launch, step 0 reads the
 * tensor itself (a replayed tape needs no initial mask). */
int tap_stepper_begin(tap_stepper *s, const float *static_, const float *dyn_in, int flags, void *stream);

/* The same for a window whose bit shadow the caller already holds (tap_rolling_window / tap_rolling_step emit it
 * next to the tensor): step 0 reads `bits` (B, nR) -- not phase 0's buffer -- and starts from an all-ones mask.
 * No launch. */
int tap_stepper_begin_shadow(tap_stepper *s, const float *static_, const unsigned long long *bits, int flags);

/* One decoding step (model.py:376-465 in one launch, or the two launches tap_transition_bits runs for the shapes
 * without a single kernel): ptr (B,) int64.  Writes phase (k & 1) of bits / dyn / current / mask, plus feature,
 * decoder_static, tour[:, k] and -- at the last step -- ratio.  TAP_E_STEPS after `steps` steps. */
int tap_stepper_step(tap_stepper *s, const int64_t *ptr, void *stream);
/* steps taken since tap_stepper_begin */
int tap_stepper_steps_done(const tap_stepper *s);

/* The same for rolling.validate's loop (rolling.py:589-637; tap_rolling_window / tap_rolling_step): the caller
 * allocates two phases of the window's static tensor and node list (a step reads the current window's and writes
 * the next one's), one buffer of everything else; a step is one call with two arguments and its launch also
 * writes decoder_static, the tour column and the picked block's GLOBAL id (sub_graph_nodes[ptr], rolling.py:637).
 * After tap_roller_begin the first window is in phase 0; after step k the next window is in phase (k + 1) & 1.
 * After N - child steps the current window is the instance's last graph: run its episode with a tap_stepper
 * (tap_stepper_begin_shadow on static_[phase] and bits, TAP_SB_CONTINUE, tour_col0 = N - child). */
typedef struct tap_roller tap_roller;

typedef struct tap_roller_buffers {
    float *static_[2];              /* (B, 1+D, child*R) */
    int32_t *nodes[2];              /* (B, child): sorted global block ids = static's columns */
    float *dynamic;                 /* (B, 3*child, child*R); nullable when bits is given: no fp32 expansion */
    unsigned long long *bits;       /* (B, child*R) nullable: dynamic's bit shadow (needs 3*child <= 64) */
    float *colsum;                  /* (B, 3, child*R) nullable */
    float *current_mask;            /* (B, child*R) nullable (model.py:297-307) */
    int32_t *err;                   /* (B,) nullable: raised to 1 when a window could not be filled; cleared by begin */
    float *feature;                 /* (B, feature_len) nullable */
    float *decoder_static;          /* (B, D) nullable */
    int64_t *tour;                  /* (B, tour_stride) nullable: column k = the pick in window k */
    int32_t *picked;                /* (B, tour_stride) nullable: column k = global id of the block picked in window k */
    int32_t tour_stride;            /* >= N - child (N to leave room for the last window's episode) */
} tap_roller_buffers;

int tap_roller_create(tap_ctx *ctx, const tap_env_desc *d, void *env_state, int N, int child,
                      const tap_roller_buffers *buf, tap_roller **out);
void tap_roller_destroy(tap_roller *r);
/* blocks / rel / state as tap_rolling_init produced them (state is consumed: re-run tap_rolling_init, or restore
 * a saved copy, per episode).  Emits the first window into phase 0. */
int tap_roller_begin(tap_roller *r, const int32_t *blocks, const uint64_t *rel, uint64_t *state, void *stream);
/* place the block picked in the current window + emit the next window: tap_rolling_step */
int tap_roller_step(tap_roller *r, const int64_t *ptr, void *stream);
int tap_roller_steps_done(const tap_roller *r);

/* ---- measurement support ------------------------------------------------------------------ */

/* Bandwidth calibration in the hot kernels' own access shape (16 bytes per lane, a wavefront covers 1 KiB): SURVEY
 * 8(d) asks for the HBM peak to be confirmed on the box.  kind 0: dst = src (plain stores), 1: the same with
 * nontemporal stores, 2: fill dst (plain), 3: fill dst (nontemporal), 4: read src only, 5: fill with write-through
 * (sc0 sc1) stores, 6 / 7: the same in the bit-shadow expansion's store shape at c2 and as 4 800 linear bytes per wave
 * (bytes % 19200 == 0), 8: the same bytes as 960-byte store instructions that start on 64 bytes.  bytes % 16 == 0,
 * 16-byte aligned buffers.  scripts/calibrate_bw.py times these.  kinds 9 .. 13: SPARSE reads of src (256-byte aligned,
 * bytes % 256 == 0), one record per stride, every record once -- 9: 8 bytes of every 128, 10: 32 of 128, 11: 32 of 64,
 * 12: 64 of 128, 13: 32 of 256 -- for scripts/calibrate_fetch.py, which reads rocprofv3's FETCH_SIZE against them.
 * Nothing on the hot path calls these. */
int tap_bw_probe(tap_ctx *ctx, int kind, void *dst, const void *src, size_t bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TAPENV_H */
