// macs.hip -- MACS / MUL 2D placement (tools.py:2456-2749).  Placeholder until the kernel lands.
#include "tap_common.h"

int tap_macs2d_step(tap_ctx *ctx, const StepArgs &, hipStream_t)
{
    return tap_fail(ctx, TAP_E_UNSUPPORTED, "MACS/MUL placement is not implemented yet");
}
