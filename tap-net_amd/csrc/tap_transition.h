// tap_transition.h -- what the fused-step kernels of transition.hip share (kept apart so that a further translation
// unit of fused kernels can compile beside it): the argument
// block, the stream wave (the precedence update of one or two envs by one wavefront) and the workgroup geometry.
#pragma once

#include "tap_common.h"
#include "tap_masks.h"
#include "tap_place.h"

// -DTAP_PROF / -DTAP_PROF_SWITCH builds only: flag bits (beside TAP_T_*) that switch one kind of wave off (scripts/decompose_step.py)
constexpr int TAP_T_PROF_NOSTREAM = 1 << 8, TAP_T_PROF_NOPLACE = 1 << 9;

struct TransArgs {
    StepArgs s;   // placement (always the gather form: s.static_, s.ptr)
    MaskArgs m;   // precedence update
    int flags;
    float *ratio_out;
};

// ---- a stream wave: out-of-place copy of SPW consecutive slabs with the chosen rows cleared
//      (pack.py:370-374), then the column sums + both masks (pack.py:318-329)
// MODE & 3: 0 = fp32 copy with the column-sum shadow, 1 = on the bit shadow, 2 = first step (shadow built in the launch);
// MODE & 4 (TAP_MODE_MERGED): the fp32 expansion walks the wave's two slabs as one run of rows (tap_masks.h:
// stream_wave_bits, where the A/B figures are) instead of slab by slab
// MODE & 8 / & 16 (TAP_MODE_C4_5 / _15): the window is the reference's own -- n = 10, rows = 30, nR = 20 (2D) / 60 (3D) --
// and its shape is compiled in (stream_wave_bits_r4: C4S); the launcher checks the shape
// (both bits, TAP_MODE_C4_10: c4's window, n = 20, rows = 60, nR = 40 -- the MACS 2D step, transition_macs.hip)
// MODE & 32 (TAP_MODE_INPLACE, with MODE & 3 == 1 only): dyn_out holds the previous step's tensor (MaskArgs::inplace) --
// the stream waves write the cleared rows' zeros instead of expanding the slab
// MODE & 64 (TAP_MODE_FULL, with MODE & 3 == 1 and a compiled-in shape only): ptr, static and mask_in are all given and B is
// a multiple of the workgroup's envs -- the stream wave carries no code for absent inputs or idle slabs (tap_masks.h: FULL)
constexpr int TAP_MODE_MERGED = 4, TAP_MODE_C4_5 = 8, TAP_MODE_C4_15 = 16, TAP_MODE_C4_10 = 24, TAP_MODE_INPLACE = 32, TAP_MODE_FULL = 64;
__host__ __device__ constexpr int tap_mode_shape(int D) { return D == 2 ? TAP_MODE_C4_5 : TAP_MODE_C4_15; }
inline bool tap_mode_shape_ok(const MaskArgs &m, int D) { return m.n == 10 && m.rows == 30 && m.update_rows == 3 && m.nR == (D == 2 ? 20 : 60); }
inline bool tap_mode_shape20_ok(const MaskArgs &m) { return m.n == 20 && m.rows == 60 && m.update_rows == 3 && m.nR == 40; }
template <int SPW, int NC, int MODE_>
__device__ __forceinline__ void trans_stream_wave(const MaskArgs &m, int senv0, int lane, float *lds)
{
    constexpr int MODE = MODE_ & 3;
    constexpr bool MERGED = (MODE_ & TAP_MODE_MERGED) != 0;
    constexpr bool INPLACE = (MODE_ & TAP_MODE_INPLACE) != 0, FULL = (MODE_ & TAP_MODE_FULL) != 0;
    static_assert(!INPLACE || MODE == 1, "in place: only on a shadow the caller hands in");
    constexpr int C4S = (MODE_ & TAP_MODE_C4_10) == TAP_MODE_C4_10 ? 10 : (MODE_ & TAP_MODE_C4_5) ? 5 : (MODE_ & TAP_MODE_C4_15) ? 15 : 0;
    static_assert(C4S == 0 || NC == 1, "the compiled-in window shapes have one column per lane");
    bool on[SPW];
#pragma unroll
    for (int k = 0; k < SPW; ++k) on[k] = senv0 + k < m.B;
    TL_STAMP(0);
    if (NC > 0) {
        if (MODE == 1) stream_wave_bits<SPW, (NC > 0 ? NC : 1), false, MERGED, C4S, INPLACE, FULL>(m, senv0, lane, on, lds);
        else if (MODE == 2) stream_wave_bits<SPW, (NC > 0 ? NC : 1), true, MERGED, C4S>(m, senv0, lane, on, lds);
        else stream_wave_fast<SPW, (NC > 2 ? 4 : 6), (NC > 0 ? NC : 1)>(m, senv0, lane, on, lds);
        TL_STAMP(2);
        TL_WAIT_VM();
        TL_STAMP(3);
        return;
    }
    const size_t slab = (size_t)m.rows * m.nR;
#pragma unroll
    for (int k = 0; k < SPW; ++k) {
        if (!on[k]) continue;
        const int senv = senv0 + k;
        bool badp;
        const long pc = tap_col((long)m.ptr[senv], m.nR, badp);
        // pack.py:339; an index outside [0, nR) clears nothing and removes no column (tap_masks.h)
        const long real = badp ? -1 : (long)m.static_[(size_t)senv * m.static_rows * m.nR + pc];
        const long p = badp ? -1 : pc;
        const ClearRanges cr = clear_ranges(m, real);
        const float *src = m.dyn_in + (size_t)senv * slab;
        float *dst = m.dyn_out + (size_t)senv * slab;
        for (long f = lane; f < (long)slab; f += 64) {
            float v = src[f];
            if (in_cleared(cr, (int)f)) v = 0.f;
            dst[f] = v;
        }
        mask_env(m, senv, lane, real, p);
    }
}

template <int G, int SW> struct TransGeom {
    static constexpr int EPB = (G == 64) ? 4 : 8;   // envs per workgroup
    static constexpr int ENV_WAVES = EPB * G / 64;  // waves made of placement lane groups
    static constexpr int STREAM_WAVES = (SW < EPB) ? SW : EPB; // waves that stream the dynamic slabs
    static constexpr int SPW = EPB / STREAM_WAVES;  // slabs per stream wave
    static constexpr int THREADS = 64 * (ENV_WAVES + STREAM_WAVES);
};

