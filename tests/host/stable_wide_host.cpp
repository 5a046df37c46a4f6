// Host build of tap-net_amd/csrc/tap_stable_wide.h (tools.is_stable for footprints up to 16 x 16, the form the
// big-container kernels call for block sides above 8) behind a C entry point, for tests/test_stable_wide_host_cpu.py:
// mask = one byte per footprint cell, row-major (i outer, j inner), 1 = supported.
#include <cstdint>

#include "tap_stable_wide.h"

extern "C" int sw_is_stable(int bx, int by, const uint8_t *mask)
{
    // a two-level height-map: supported cells at the resting level z = 1, the others below it
    return tap_stable3d_wide([&](int i, int j) { return mask[i * by + j] ? 1 : 0; }, bx, by, 1);
}
