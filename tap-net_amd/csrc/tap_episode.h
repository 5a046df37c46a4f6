// tap_episode.h -- what the whole-episode kernels share (episode.hip: the lane-per-cell forms; big.hip: one wavefront
// per container above 64 cells): the argument block, one tour entry's block, and the scores of the final state.
#pragma once

#include "tap_common.h"
#include "tap_place.h"

struct EpisodeArgs {
    tap_env_desc d;
    int B, n;
    const float *static_;
    int static_rows, nR;
    const int64_t *tour;
    const int32_t *blocks; // (B, n, D) explicit block lists when static_ is null (tap_pack_blocks)
    int target_sel;        // 0 | 1: only the tour entries whose target id (last row of static_) equals it; -1: all
    const uint32_t *lut;
    float *reward_out;
    int32_t *pos_out;
    uint8_t *stable_out;
    double *score64_out;   // the function's `ratio` as fp64, 0 for an empty list (pack.py:459-470, 760-769)
    int64_t *scores_out;   // (B, 5): valid_size, box_size, empty_size, stable_num, max(heightmap)
    int32_t *err_out;      // (B,) the sticky error bits of tap_env_check
};

// One tour entry: the block's sides and whether it belongs to this container's list.
// pack.py:441-444 gather by tour, :454-455 rows 1..D, tools.py:2415 / :3249 astype('int'); pack.py:455-457,
// 755-757: with two containers the list of one is the sub-sequence with its target id.
template <int D>
__device__ __forceinline__ bool episode_block(const EpisodeArgs &a, int env, int t, bool ev, int (&dims)[3], int &err)
{
    dims[0] = dims[1] = dims[2] = 1;
    if (!ev) return false;
    if (a.static_) {
        bool badp;
        const long p = tap_col((long)a.tour[(size_t)env * a.n + t], a.nR, badp);
        if (badp) err |= 4;                                   // the reference's gather raises
        for (int k = 0; k < D; ++k) {
            const float v = a.static_[((size_t)env * a.static_rows + 1 + k) * a.nR + p];
            dims[k] = badp ? 0 : (int)v;
        }
        if (a.target_sel >= 0) {
            const float id = a.static_[((size_t)env * a.static_rows + (a.static_rows - 1)) * a.nR + p];
            if (badp || id != (float)a.target_sel) return false;
        }
        return !badp;
    }
    bool in = true;
    for (int k = 0; k < D; ++k) {
        dims[k] = a.blocks[((size_t)env * a.n + t) * D + k];
        in = in && dims[k] >= 1;                              // a side < 1 marks "not in this list"
    }
    return in;
}

// `ratio` of calc_positions_lb_greedy / calc_positions_mcs (tools.py:2442-2445, 3279-3308): Container.calc_ratio's
// table without the division; 'C+P-lb-soft' is C + P + S here (tools.py:2442).
__device__ __forceinline__ double episode_ratio(int mode, double C, double P, double S)
{
    switch (mode) {
    case TAP_R_C: return C;
    case TAP_R_CxS: return C * S;
    case TAP_R_CP: return C + P;
    case TAP_R_CPxS: return (C + P) * S;
    case TAP_R_2CPS: return (2 * C + P) + S;
    case TAP_R_CxPxS: return (C * P) * S;
    default: return (C + P) + S;
    }
}

// tools.py:2434-2448 / 3265-3312 on the final state; one lane per container calls this
__device__ __forceinline__ void episode_finish(const EpisodeArgs &a, int env, const Counters &cnt, int gmax, int err)
{
    const long long box = (long long)gmax * a.d.W * a.d.L;
    const double C = (double)cnt.valid / (double)box;
    const double P = (double)cnt.valid / (double)((long long)cnt.empty + cnt.valid);
    // S over blocks_num = len(blocks): the entries of this container's list, failed placements included
    const double S = (double)cnt.nstable / (double)cnt.count;
    const double score = cnt.count ? episode_ratio(a.d.ratio_mode, C, P, S) : 0.0;   // pack.py:459-466, 760-769
    // the reference raises on a height overflow / bad index; here the score becomes NaN and err_out says why
    if (a.reward_out) a.reward_out[env] = err ? __int_as_float(0x7fc00000) : -(float)score;
    if (a.score64_out) a.score64_out[env] = err ? __longlong_as_double(0x7ff8000000000000ll) : score;
    if (a.scores_out) {
        int64_t *s = a.scores_out + (size_t)env * 5;
        s[0] = cnt.valid; s[1] = cnt.count ? box : 0; s[2] = cnt.empty; s[3] = cnt.nstable; s[4] = gmax;
    }
    if (a.err_out) a.err_out[env] = err;
}

