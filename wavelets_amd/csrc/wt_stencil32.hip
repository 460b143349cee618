// Translation unit of the float32 per-scale kernels: wt_stencil.h instantiated for float (chain, lattice and
// row kernels in every mode).  gfx950 only.  A unit of its own (round 5) because its code object is the
// library's largest (5 MB: ~110 instantiations) and the runtime loads a unit's device code on the first launch
// from it (~10 ms): a process whose calls stay on the fused passes (decompose / denoise of the built-in
// families) no longer pays for kernels it never launches.  Compiled with -DWT_TU_NAME=stencil32.
#include <hip/hip_runtime.h>

#include "wt_internal.h"
#include "wt_stencil_launch.h"
#include "wt_unit_probe.h"

WT_UNIT_PROBE_DEFINE

int wt32_stencil_launch(const StencilCtx &sc, int mode, const ChainArgs &a, int s, const char *name)
{
    switch (mode) {
        case MODE_SMOOTH: return wt_launch_stencil<float, MODE_SMOOTH>(sc, a, s, name);
        case MODE_SMOOTH_SQ: return wt_launch_stencil<float, MODE_SMOOTH_SQ>(sc, a, s, name);
        case MODE_DECOMP: return wt_launch_stencil<float, MODE_DECOMP>(sc, a, s, name);
        case MODE_VAR: return wt_launch_stencil<float, MODE_VAR>(sc, a, s, name);
        case MODE_WOW: return wt_launch_stencil<float, MODE_WOW>(sc, a, s, name);
        case MODE_WOW_PLAIN: return wt_launch_stencil<float, MODE_WOW_PLAIN>(sc, a, s, name);
        case MODE_WOW_GAMMA: return wt_launch_stencil<float, MODE_WOW_GAMMA>(sc, a, s, name);
    }
    WT_FAIL("float32 plan: unknown stencil mode %d", mode);
}
