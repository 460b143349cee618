// Host side of the per-scale stencil kernels (wt_stencil.h): which kernel serves a dilation and how its
// grid is cut, written once for both element types.  Included by wt_transform.hip (T = float) and by
// wt_stencil64.hip (T = double).
#pragma once
#include <algorithm>
#include <cstdlib>

#include "wt_internal.h"
#include "wt_stencil.h"

// tuning / A-B switches (wt_set_option; defined in wt_transform.hip)
extern int g_opt_row_kernel;
extern int g_opt_lattice;

// what a launch needs to know about the plan it runs on
struct StencilCtx {
    wt_ctx *ctx;
    hipStream_t stream;   // launch stream (the context's compute stream, or its side stream)
    Geo g;                // P in elements of the plan's type
    int family;           // WT_B3SPLINE / WT_TRIANGLE
};

static inline int wt_family_taps(int family) { return family == WT_B3SPLINE ? 5 : 3; }

// chunking of the polyphase chains: enough (phase, chunk) items to fill the chip, chunks long
// enough that the K-1 warm-up rows stay a small fraction.  gx = workgroups (of 64 x 4 lanes) along x.
template <typename T>
static int wt_chain_geometry(const Geo &g, int s, ChainArgsT<T> &a, dim3 &grid, dim3 &block, int gx_override = 0)
{
    constexpr int PX = WtVec<T>::PX;
    const int d = 1 << s;
    const int X = (g.W + PX - 1) / PX;           // 16-byte groups per row
    const int gx = gx_override ? gx_override : (X + 63) / 64;
    const int n_max = (g.nrows + d - 1) / d;     // longest chain
    static const int64_t lanes_env = getenv("WT_CHAIN_LANES") ? atoll(getenv("WT_CHAIN_LANES")) : 0;   // experiments
    static const int smax_env = getenv("WT_CHAIN_SMAX") ? atoi(getenv("WT_CHAIN_SMAX")) : 0;
    const int64_t want_items = std::max<int64_t>(1, (lanes_env > 0 ? lanes_env : (int64_t)524288) / std::max(1, gx * 64));
    int chunks_target = (int)std::max<int64_t>(1, want_items / std::min(d, g.nrows));
    int S = (n_max + chunks_target - 1) / chunks_target;
    S = std::max(S, std::min(n_max, 8));
    S = std::min(S, smax_env > 0 ? smax_env : 64);
    int chunks = (n_max + S - 1) / S;
    int64_t items = (int64_t)d * chunks;
    while ((items + 3) / 4 > 65528) {             // grid.y limit
        S *= 2;
        chunks = (n_max + S - 1) / S;
        items = (int64_t)d * chunks;
    }
    a.g = g;
    a.d = d;
    a.S = S;
    a.chunks = chunks;
    grid = dim3(gx, (unsigned)(((items + 3) / 4 + 7) / 8 * 8));   // multiple of 8: wt_xcd_remap
    block = dim3(64, 4);
    return 0;
}

// Geometry of the bilateral marches (wt_bilateral2_kernel, wt64_bilateral_march_kernel): NW waves side by side on one
// chain item (phase x chunk), `wx` waves across a row; the chunking is wt_chain_geometry's, the items run over grid.y.
template <typename T>
static int wt_march_geometry(const Geo &g, int s, ChainArgsT<T> &a, dim3 &grid, dim3 &block, int wx, int NW)
{
    WT_TRY(wt_chain_geometry<T>(g, s, a, grid, block, wx));
    int64_t items = (int64_t)a.d * a.chunks;
    while (items > 65528) {                               // grid.y limit
        a.S *= 2;
        a.chunks = ((g.nrows + a.d - 1) / a.d + a.S - 1) / a.S;
        items = (int64_t)a.d * a.chunks;
    }
    grid = dim3((wx + NW - 1) / NW, (unsigned)((items + 7) / 8 * 8));   // multiple of 8: wt_xcd_remap
    block = dim3(64, NW);
    return 0;
}

// Row kernel (taps from an LDS copy of the row) where the horizontal halo fits the workgroup.
template <typename T, int K, int MODE, int NW>
static int wt_launch_row_t(const StencilCtx &sc, ChainArgsT<T> a, int HX, const char *name)
{
    constexpr int PX = WtVec<T>::PX;
    constexpr int NL = NW * 64;
    const Geo &g = sc.g;
    const int d = a.d;
    const int VXMAX = (NL * PX - 2 * HX) / 32 * 32;
    const int W4 = (g.W + PX - 1) / PX * PX;
    const int nx = (W4 + VXMAX - 1) / VXMAX;
    RowArgsT<T> ra{};
    ra.HX = HX;
    ra.Vx = std::min(VXMAX, ((W4 + nx - 1) / nx + 31) / 32 * 32);
    const int phases = std::min(d, g.nrows);
    const int n_max = (g.nrows + d - 1) / d;
    // one to two rounds of resident workgroups (16 waves per CU at <= 128 VGPRs)
    const int slots = sc.ctx->num_cus * (16 / NW) * 2;
    int chunks = std::max(1, slots / std::max(1, nx * phases));
    int S = (n_max + chunks - 1) / chunks;
    S = std::max(S, std::min(n_max, 16));
    chunks = (n_max + S - 1) / S;
    a.S = S;
    a.chunks = chunks;
    ra.c = a;
    const int64_t gy = (int64_t)d * chunks;
    if (gy > 65535) WT_FAIL("row kernel: grid too large");
    dim3 grid(nx, (unsigned)gy), block(NL);
    ProfScope ps(sc.ctx, name, sc.stream);
    if (d < PX) hipLaunchKernelGGL((wt_row_kernel<T, K, MODE, true, NW>), grid, block, 0, sc.stream, ra);
    else hipLaunchKernelGGL((wt_row_kernel<T, K, MODE, false, NW>), grid, block, 0, sc.stream, ra);
    WT_HIP(hipGetLastError());
    return 0;
}

template <typename T>
static const char *wt_row_name(int mode)
{
    constexpr bool f64 = sizeof(T) == 8;
    switch (mode) {
        case MODE_SMOOTH: return f64 ? "wt64_row_kernel<smooth>" : "wt_row_kernel<smooth>";
        case MODE_SMOOTH_SQ: return f64 ? "wt64_row_kernel<smooth_sq>" : "wt_row_kernel<smooth_sq>";
        case MODE_DECOMP: return f64 ? "wt64_row_kernel<decomp>" : "wt_row_kernel<decomp>";
        case MODE_VAR: return f64 ? "wt64_row_kernel<variance>" : "wt_row_kernel<variance>";
        default: return f64 ? "wt64_row_kernel<wow>" : "wt_row_kernel<wow>";
    }
}

template <typename T>
static const char *wt_lattice_name(int mode)
{
    constexpr bool f64 = sizeof(T) == 8;
    switch (mode) {
        case MODE_SMOOTH: return f64 ? "wt64_lattice_kernel<smooth>" : "wt_lattice_kernel<smooth>";
        case MODE_SMOOTH_SQ: return f64 ? "wt64_lattice_kernel<smooth_sq>" : "wt_lattice_kernel<smooth_sq>";
        case MODE_DECOMP: return f64 ? "wt64_lattice_kernel<decomp>" : "wt_lattice_kernel<decomp>";
        case MODE_VAR: return f64 ? "wt64_lattice_kernel<var>" : "wt_lattice_kernel<var>";
        default: return f64 ? "wt64_lattice_kernel<wow>" : "wt_lattice_kernel<wow>";
    }
}

// One scale of a built-in family: lattice kernel for the large dilations, row kernel where the x halo
// fits a workgroup, the chain kernel otherwise.  `name`: profiler name of the chain kernel.
template <typename T, int MODE>
static int wt_launch_stencil(const StencilCtx &sc, ChainArgsT<T> a, int s, const char *name)
{
    constexpr int PX = WtVec<T>::PX;
    const bool no_row = !g_opt_row_kernel;
    const Geo &g = sc.g;
    const int hw = wt_family_taps(sc.family) / 2;
    const int d = 1 << s;
    const int HX = std::max(32, (hw * d + 31) / 32 * 32);
    const bool b3 = sc.family == WT_B3SPLINE;
    a.g = g;
    a.d = d;
    a.nt = (int64_t)g.nrows * g.P * (int64_t)sizeof(T) >= ((int64_t)32 << 20);   // planes >> L2: streaming stores
    dim3 grid, block;
    // d >= 64: lattice kernel (C lattice columns per thread share their taps); measured faster
    // than the 8-wave row kernel from d = 64 up and 2.6x faster than the chain kernel at d >= 256
    static const int lat_min_d = getenv("WT_LATTICE_MIN_D") ? std::max(4, atoi(getenv("WT_LATTICE_MIN_D"))) : 64;
    const int lat_c = (g_opt_lattice && d >= lat_min_d && g.border == 0 && g.W % PX == 0) ? (g.W >= 4 * d ? 4 : (g.W >= 2 * d ? 2 : 0)) : 0;
    if (lat_c) {
        const int J = (g.W + d - 1) / d;                          // lattice columns per phase
        const int tx = ((J + lat_c - 1) / lat_c) * (d / PX);      // threads along x
        WT_TRY(wt_chain_geometry<T>(g, s, a, grid, block, (tx + 63) / 64));
        ProfScope ps(sc.ctx, wt_lattice_name<T>(MODE), sc.stream);
        if (b3 && lat_c == 4) hipLaunchKernelGGL((wt_lattice_kernel<T, 5, MODE, 4>), grid, block, 0, sc.stream, a);
        else if (b3) hipLaunchKernelGGL((wt_lattice_kernel<T, 5, MODE, 2>), grid, block, 0, sc.stream, a);
        else if (lat_c == 4) hipLaunchKernelGGL((wt_lattice_kernel<T, 3, MODE, 4>), grid, block, 0, sc.stream, a);
        else hipLaunchKernelGGL((wt_lattice_kernel<T, 3, MODE, 2>), grid, block, 0, sc.stream, a);
        WT_HIP(hipGetLastError());
        return 0;
    }
    // (halo lanes on both sides of a workgroup of NW * 64 lanes of PX pixels)
    if (!no_row && 2 * HX <= 64 * PX) {
        return b3 ? wt_launch_row_t<T, 5, MODE, 4>(sc, a, HX, wt_row_name<T>(MODE)) : wt_launch_row_t<T, 3, MODE, 4>(sc, a, HX, wt_row_name<T>(MODE));
    }
    if (!no_row && 2 * HX <= 128 * PX) {
        return b3 ? wt_launch_row_t<T, 5, MODE, 8>(sc, a, HX, wt_row_name<T>(MODE)) : wt_launch_row_t<T, 3, MODE, 8>(sc, a, HX, wt_row_name<T>(MODE));
    }
    WT_TRY(wt_chain_geometry<T>(g, s, a, grid, block));
    ProfScope ps(sc.ctx, name, sc.stream);
    const bool small = a.d < PX;
    if (b3 && small) hipLaunchKernelGGL((wt_chain_kernel<T, 5, MODE, true>), grid, block, 0, sc.stream, a);
    else if (b3) hipLaunchKernelGGL((wt_chain_kernel<T, 5, MODE, false>), grid, block, 0, sc.stream, a);
    else if (small) hipLaunchKernelGGL((wt_chain_kernel<T, 3, MODE, true>), grid, block, 0, sc.stream, a);
    else hipLaunchKernelGGL((wt_chain_kernel<T, 3, MODE, false>), grid, block, 0, sc.stream, a);
    WT_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// entry points of wt_stencil32.hip / wt_stencil64.hip (translation units of their own: ~110 kernel
// instantiations each)
// ---------------------------------------------------------------------------------------------
// one scale in `mode` (MODE_*) on float planes; `name`: profiler name of the chain kernel
int wt32_stencil_launch(const StencilCtx &sc, int mode, const ChainArgsT<float> &a, int s, const char *name);
// one scale in `mode` (MODE_*) on double planes
int wt64_stencil_launch(const StencilCtx &sc, int mode, const ChainArgsT<double> &a, int s);
// the range-weighted dilated filter of one scale (watroo/wavelets.py:74-105) on double planes: the
// marching kernel of wt_bilateral64.h; var == nullptr: variance of wavelets.py:434-436 formed in the kernel
int wt64_bilateral_launch(const StencilCtx &sc, const double *in, const double *var, double *out, double *out_w, int s,
                          double f1, double f2);
