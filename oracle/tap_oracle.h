/*
 * tap_oracle.h -- CPU oracle for the Transport-and-Pack environment hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C, single-threaded-per-env, voxel-level
 * restatement of the reference's (Juzhan/TAP-Net) Python algorithm.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * (tap-net_amd/, libtapenv.so) never links, imports or calls anything in oracle/.
 *
 * Parity status: PINNED.  The oracle is checked (tests/test_oracle_golden.py) against
 * golden vectors produced by importing the reference itself in the build container
 * (tests/golden/make_golden.py, outputs committed under tests/golden/), and -- where
 * /root/reference exists -- live against the reference (tests/test_oracle_vs_reference.py).
 *
 * Every function cites the reference file:line it restates.  Unlike the HIP kernels
 * (which carry only the height-map), the oracle keeps the reference's full state: the
 * voxel grid `container`, the height-map, the placed-block history and MACS's
 * per-level free-space lists, and manipulates them the way the Python does.
 */
#ifndef TAP_ORACLE_H
#define TAP_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* packing strategy (tools.py:3617-3620, 3679-3690) */
enum { ORC_LB_GREEDY = 0, ORC_MACS = 1 /* 'MACS' and 'MUL' run the same function: tools.py:2451 (2D), :2751 (3D) */,
       ORC_LB = 2 /* legacy 'LB': calc_one_position_greedy, tools.py:1602-1955 */ };

/* flag word = the string tests the reference performs on reward_type */
enum {
    ORC_F_HARD     = 1 << 0, /* reward_type.endswith('hard')   tools.py:2113,2580 */
    ORC_F_USE_P    = 1 << 1, /* 'P' in reward_type              tools.py:2135,2600 */
    ORC_F_USE_S    = 1 << 2, /* 'S' in reward_type              tools.py:2138,2602 */
    ORC_F_MCS_ZERO = 1 << 3, /* reward_type.startswith('mcs')   tools.py:2709      */
    ORC_F_MCS_TIE  = 1 << 4  /* 'mcs' in reward_type            tools.py:2718      */
};

/* Container.calc_ratio formula selector (tools.py:3907-3966) */
enum {
    ORC_R_C = 0,       /* 'comp'                          -> C / 3          */
    ORC_R_CxS,         /* 'soft','hard'                   -> C*S / 3        */
    ORC_R_CP,          /* 'pyrm'                          -> (C+P) / 3      */
    ORC_R_CPxS,        /* 'pyrm-soft/hard','mcs-soft/hard'-> (C+P)*S / 3    */
    ORC_R_CPS,         /* all 'C+P(+S)-*' types etc.      -> (C+P+S) / 3    */
    ORC_R_2CPS,        /* 'pyrm-*-SUM'                    -> (2C+P+S) / 3   */
    ORC_R_CxPxS,       /* 'CPS'                           -> C*P*S / 3      */
    ORC_R_CP_HALF      /* 'C+P-lb-soft'                   -> (C+P) / 2      tools.py:3961-3962 */
};

/* feature returned by add_new_block (tools.py:3716-3744) */
enum { ORC_FEAT_FULL = 0, ORC_FEAT_ZERO = 1, ORC_FEAT_DIFF = 2 };

/* error codes (negative) */
enum {
    ORC_OK = 0,
    ORC_E_ARG = -1,        /* bad argument (dim, size < 1, too many blocks) */
    ORC_E_HEIGHT = -2,     /* placement reaches above H: the reference raises IndexError
                              (tools.py:2109) or silently clips voxels (tools.py:2169) */
    ORC_E_REF_RAISES = -3, /* the reference would raise a Python exception here */
    ORC_E_CAP = -4         /* internal capacity exceeded */
};

typedef struct {
    int32_t D;          /* 2 | 3 */
    int32_t W, L, H;    /* container_size; L = 1 when D == 2 */
    int32_t n_max;      /* blocks_num */
    int32_t strategy;   /* ORC_LB_GREEDY | ORC_MACS */
    int32_t flags;      /* ORC_F_* */
    int32_t ratio_mode; /* ORC_R_* */
    int32_t feature;    /* ORC_FEAT_* */
} orc_desc;

typedef struct orc_env orc_env;

/* tools.Container.__init__ (tools.py:3611-3661) */
orc_env *orc_env_new(const orc_desc *d);
void     orc_env_free(orc_env *e);
/* tools.Container.clear_container (tools.py:3858-3885) */
void     orc_env_clear(orc_env *e);
/* tools.Container.add_new_block (tools.py:3663-3744).  `block` = D ints.
 * feature_out (nullable): FULL/ZERO -> W*L ints; DIFF -> W-1 ints (2D) or 2*W*L ints (3D). */
int      orc_env_add_block(orc_env *e, const int32_t *block, int32_t *feature_out);
/* tools.Container.get_heightmap (tools.py:3824-3856) */
void     orc_env_feature(const orc_env *e, int32_t *feature_out);
/* tools.Container.calc_CPS / calc_ratio (tools.py:3887-3966) */
void     orc_env_cps(const orc_env *e, double cps[3]);
double   orc_env_ratio(const orc_env *e);

/* state accessors */
const int32_t *orc_env_heightmap(const orc_env *e); /* W*L, x-major */
const int32_t *orc_env_positions(const orc_env *e); /* n_max*D */
const uint8_t *orc_env_stable(const orc_env *e);    /* n_max */
const int32_t *orc_env_container(const orc_env *e); /* W*L*H voxels, numpy (W,(L),H) order */
int64_t  orc_env_valid(const orc_env *e);
int64_t  orc_env_empty(const orc_env *e);
int32_t  orc_env_count(const orc_env *e);           /* current_blocks_num */
int32_t  orc_env_error(const orc_env *e);           /* sticky first error */
int32_t  orc_feature_len(const orc_desc *d);

/* predicates, exposed for the exhaustive pin tests */
/* tools.is_stable_2d (tools.py:839-868); support[i] != 0 means supported */
int orc_is_stable_2d(const int32_t *support, int obj_left, int obj_width);
/* tools.is_stable (tools.py:710-765) on a support mask: mask[i*by+j] != 0 iff the voxel under
 * footprint cell (i,j) is > 0.  z == 0 is handled by the caller. */
int orc_is_stable_3d_mask(int bx, int by, const uint8_t *mask);

/* ---- batched drivers (loops over envs; nthreads > 1 uses OpenMP when compiled in) ---- */

/* B independent episodes from empty containers: blocks (B,n,D) int32 in placement order.
 * Outputs (all nullable): positions (B,n,D), stable (B,n), features (B,n,feat_len),
 * heightmaps (B,n,W*L) after every step, ratio64 (B,) = Container.calc_ratio(),
 * cps (B,3), counters (B,3) = valid, empty, current_blocks_num, errs (B,).
 * Returns the number of envs that hit an error. */
int orc_run_episodes(const orc_desc *d, int B, int n, const int32_t *blocks,
                     int32_t *positions, uint8_t *stable, int32_t *features,
                     int32_t *heightmaps, double *ratio64, double *cps, int64_t *counters,
                     int32_t *errs, int nthreads);

/* tools.calc_positions_lb_greedy (tools.py:2393-2449): one episode, returns the UN-normalised
 * ratio C+P+S and scores = [valid, box, empty, stable_num, max_h]. */
int orc_calc_positions_lb_greedy(const orc_desc *d, int n, const int32_t *blocks,
                                 int32_t *positions, uint8_t *stable, double *ratio,
                                 int64_t scores[5]);

/* tools.calc_positions_mcs (tools.py:3213-3315): one episode with the MACS / MUL per-block function; `ratio` by
 * the reward type (d->ratio_mode, un-normalised), scores as above. */
int orc_calc_positions_mcs(const orc_desc *d, int n, const int32_t *blocks, int32_t *positions,
                           uint8_t *stable, double *ratio, int64_t scores[5]);

/* the figures pack.render (pack.py:670-807) writes per sample: ratio_out (B,), scores_out (B,5) fp64 =
 * valid_size, box_size, empty_size, stable_num, packing_height.  mul != 0: the two-container input types
 * (pack.py:754-790): each target id's blocks into its own container described by d_mul (3D: height =
 * initial_container_height, pack.py:720), all six figures averaged.  Returns the number of samples with an error. */
int orc_render_scores(const orc_desc *d, const orc_desc *d_mul, int B, int n, int nR, int static_rows,
                      const float *static_, const int64_t *tour, int mul, double *ratio_out,
                      double *scores_out, int32_t *errs);

/* pack.reward (pack.py:378-473): gather blocks by tour, full episode per env, -(C+P+S) as fp32.
 * static_ (B, static_rows, nR) fp32; tour (B, n) int64; reward_out (B,) fp32. */
int orc_reward(const orc_desc *d, int B, int n, int nR, int static_rows, const float *static_,
               const int64_t *tour, float *reward_out, int nthreads);

/* ---- instance generation ---- */

/* generate.calc_dependent (generate.py:575-771) on the env's voxel grid after n placements.
 * Each output is an n*n row-major 0/1 matrix; M[a*n + b] = 1 reads "block a blocks block b"
 * (move: a rests above b; left/right/forward/backward: a stands in b's side access).
 * forward/backward are all-zero in 2D.  arm_size is used in 2D only (generate.py:623-641). */
void orc_calc_dependent(const orc_env *e, int n, int arm_size, uint8_t *move, uint8_t *left,
                        uint8_t *right, uint8_t *forward, uint8_t *backward);

/* generate.generate_blocks for GIVEN block sizes (generate.py:893-971, container_width >= 0
 * branch): pack `blocks` (n, D) into the initial container with 'C+P+S-lb-hard'
 * (generate.py:908), derive the dependencies, and lay the result out as pack.PACKDataset does
 * for input_type 'bot', allow_rot=True (pack.py:101-195): static_out (1+D, n*R) fp32,
 * dynamic_out (3n, n*R) fp32, positions_out (n, D).  Returns 1 if the instance is accepted
 * (every block placed and stable, generate.py:909-910), 0 if the reference would re-sample,
 * < 0 on error. */
int orc_instance_from_blocks(int D, const int32_t *init_size, int n, int arm_size,
                             const int32_t *blocks, int32_t *positions_out, float *static_out,
                             float *dynamic_out);

/* ---- rolling windows: generate.InitialContainer (generate.py:1589-1839) ---- */

typedef struct orc_rolling orc_rolling;
/* InitialContainer.__init__: paint the initial container from blocks (N, D; rotation 0) and
 * positions (N, D), derive the five dependency graphs (calc_dependent, arm_size 1). */
orc_rolling *orc_rolling_new(int D, const int32_t *init_size, int N, int child,
                             const int32_t *blocks, const int32_t *positions);
void orc_rolling_free(orc_rolling *r);
/* InitialContainer.convert_to_input (generate.py:1778-1822) incl. sub_deps_graph (:1674-1776):
 * static_out (1+D, child*R), dynamic_out (3*child, child*R) fp32, nodes_out (child) = sorted
 * sub_graph_nodes.  Returns is_last_graph() (1 / 0), < 0 when the window could not be filled. */
int orc_rolling_window(orc_rolling *r, float *static_out, float *dynamic_out, int32_t *nodes_out);
/* InitialContainer.remove_block(sub_graph_nodes[local_index]) (generate.py:1824-1835) */
void orc_rolling_remove(orc_rolling *r, int local_index);

/* ---- PPSG: perfect-packing instances (generate.py:17-301); see the section comment in tap_oracle.c ---- */
typedef struct orc_rng orc_rng;
/* word source = an explicit 32-bit stream (words recorded from numpy's RandomState: the pin) ... */
orc_rng *orc_rng_words(const uint32_t *words, int64_t n);
/* ... or the counter generator shared with the HIP kernels */
orc_rng *orc_rng_counter(uint64_t key);
uint64_t orc_rng_key(uint64_t seed, uint64_t a, uint64_t b, uint64_t c);
void orc_rng_free(orc_rng *r);
int64_t orc_rng_consumed(const orc_rng *r);
int orc_rng_exhausted(const orc_rng *r);
/* BPP_Generator_3D (generate.py:232-301), one call: blocks, positions (n,3); -> 1 if every side is in
 * [min_size, max_size) (check_all_blocks_size, generate.py:41-53), 0 if not */
int orc_bpp3d(orc_rng *rng, int n, const int32_t *gt_size, int min_size, int max_size,
              int32_t *blocks, int32_t *positions);
/* generate.py:108-158 for a proposed layout order: hard LB_GREEDY packing, all stable, removable in reverse */
int orc_ppsg_try_layout(int n, const int32_t *init_size, int arm_size, const int32_t *blocks, int input_simple,
                        int32_t *positions_out);
/* generate_blocks_with_GT (generate.py:17-161), block_dim 3 */
int orc_generate_blocks_with_gt(orc_rng *rng, int n, const int32_t *gt_size, const int32_t *init_size,
                                int arm_size, int min_size, int max_size, int input_simple, int allow_rot,
                                int64_t max_bpp, int32_t *blocks_out, int32_t *positions_out, int64_t *stats);
/* the pieces as the HIP kernels key their counter streams (tap_ppsg_gt / tap_ppsg_order) */
int orc_ppsg_try_layout_d(int D, int n, const int32_t *init_size, int arm_size, const int32_t *blocks, int input_simple,
                          int32_t *positions_out);
/* 2D (generate.py:392-484 BPP_Generator_2D_easy; gauss: the caller's numpy-built split table, see the .c file) */
int orc_bpp2d_easy(orc_rng *rng, int n, const int32_t *gt_size, int min_size, int max_size,
                   const double *gauss, int gauss_stride, int gauss_rows, int32_t *blocks, int32_t *positions);
int orc_generate_blocks_with_gt_2d(orc_rng *rng, int n, const int32_t *gt_size, const int32_t *init_size,
                                   int arm_size, int min_size, int max_size, int input_simple, int allow_rot,
                                   int64_t max_bpp, const double *gauss, int gauss_stride, int gauss_rows,
                                   int32_t *blocks_out, int32_t *positions_out, int64_t *stats);
int64_t orc_ppsg_gt2d(uint64_t seed, int64_t instance, int gen, int n, int W, int H, int min_size, int max_size,
                      int64_t max_attempts, const double *gauss, int gauss_stride, int gauss_rows,
                      int32_t *gt_blocks, int32_t *gt_positions);
int orc_ppsg_order_2d(uint64_t seed, int64_t instance, int gen, int trial, int n, const int32_t *gt_size,
                      const int32_t *gt_blocks, const int32_t *gt_positions, int32_t *blocks_out);
int64_t orc_ppsg_gt(uint64_t seed, int64_t instance, int gen, int S, int ns, int W, const int32_t *heights,
                    int min_size, int max_size, int64_t max_attempts, int32_t *gt_blocks, int32_t *gt_positions);
int orc_ppsg_order(uint64_t seed, int64_t instance, int gen, int trial, int n, const int32_t *gt_size,
                   const int32_t *gt_blocks, const int32_t *gt_positions, int32_t *blocks_out);

/* ---- precedence tensors (pack.py:276-376, model.py:297-307) ---- */

/* OpenMP threads of the three batched drivers below (timing only; results do not depend on it) */
void orc_set_threads(int n);
/* initial mask, model.py:297-307.  dynamic (B, rows, nR) fp32, rows = 3n ('bot') or n. */
void orc_initial_mask(int B, int n, int nR, int rows, const float *dynamic, float *mask_out);
/* pack.update_dynamic, pack.py:333-376.  update_time = 1 | 3.  static_ (B, static_rows, nR). */
void orc_update_dynamic(int B, int n, int nR, int rows, int update_time, int static_rows,
                        const float *dyn_in, const float *static_, const int64_t *ptr,
                        float *dyn_out);
/* pack.update_mask, pack.py:276-331.  R = rotate_types.  Outputs: new_mask (current) and
 * chosen_mask (persistent). */
void orc_update_mask(int B, int n, int R, int rows, const float *mask_in, const float *dynamic,
                     const int64_t *ptr, float *current_out, float *mask_out);

#ifdef __cplusplus
}
#endif
#endif /* TAP_ORACLE_H */
