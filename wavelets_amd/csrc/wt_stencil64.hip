// Translation unit of the float64 per-scale kernels: wt_stencil.h instantiated for double (chain, lattice
// and row kernels in every mode) and the float64 bilateral march (wt_bilateral64.h).  gfx950 only.
// Compiled with -DWT_TU_NAME=stencil64 (wt_math64.h: per-unit names of the polynomial tables).
#include <hip/hip_runtime.h>

#include "wt_internal.h"
#include "wt_stencil_launch.h"
#include "wt_bilateral64.h"
#include "wt_unit_probe.h"

WT_UNIT_PROBE_DEFINE

int wt64_stencil_launch(const StencilCtx &sc, int mode, const ChainArgsT<double> &a, int s)
{
    if (s < 0 || s > 24) WT_FAIL("float64 plan: scale %d out of range", s);
    switch (mode) {
        case MODE_SMOOTH: return wt_launch_stencil<double, MODE_SMOOTH>(sc, a, s, "wt64_chain_kernel<smooth>");
        case MODE_SMOOTH_SQ: return wt_launch_stencil<double, MODE_SMOOTH_SQ>(sc, a, s, "wt64_chain_kernel<smooth_sq>");
        case MODE_DECOMP: return wt_launch_stencil<double, MODE_DECOMP>(sc, a, s, "wt64_chain_kernel<decomp>");
        case MODE_VAR: return wt_launch_stencil<double, MODE_VAR>(sc, a, s, "wt64_chain_kernel<variance>");
        case MODE_WOW: return wt_launch_stencil<double, MODE_WOW>(sc, a, s, "wt64_chain_kernel<wow>");
        case MODE_WOW_PLAIN: return wt_launch_stencil<double, MODE_WOW_PLAIN>(sc, a, s, "wt64_chain_kernel<wow>");
        case MODE_WOW_GAMMA: return wt_launch_stencil<double, MODE_WOW_GAMMA>(sc, a, s, "wt64_chain_kernel<wow>");
    }
    WT_FAIL("float64 plan: unknown stencil mode %d", mode);
}

int wt64_bilateral_launch(const StencilCtx &sc, const double *in, const double *var, double *out, double *out_w, int s,
                          double f1, double f2)
{
    if (s < 0 || s > 24) WT_FAIL("float64 plan: scale %d out of range", s);
    if (sc.g.border != 0 && sc.g.border != 1) WT_FAIL("bilateral kernels implement the symmetric border (whole image or polyphase) only");
    ChainArgsT<double> a{};
    a.in = in; a.out_c = out; a.out_w = out_w; a.aux = var;
    a.inline_var = var == nullptr; a.f1 = f1; a.f2 = f2;
    dim3 grid, block;
    WT_TRY(wt_march_geometry<double>(sc.g, s, a, grid, block, (sc.g.W + 63) / 64, 4));     // one pixel per lane, 4 waves side by side
    ProfScope ps(sc.ctx, "wt64_bilateral_kernel", sc.stream);
    const bool b3 = sc.family == WT_B3SPLINE;
    if (b3 && a.inline_var) hipLaunchKernelGGL((wt64_bilateral_march_kernel<5, true>), grid, block, 0, sc.stream, a);
    else if (b3) hipLaunchKernelGGL((wt64_bilateral_march_kernel<5, false>), grid, block, 0, sc.stream, a);
    else if (a.inline_var) hipLaunchKernelGGL((wt64_bilateral_march_kernel<3, true>), grid, block, 0, sc.stream, a);
    else hipLaunchKernelGGL((wt64_bilateral_march_kernel<3, false>), grid, block, 0, sc.stream, a);
    WT_HIP(hipGetLastError());
    return 0;
}
