// libwatroo_hip.so - host side of the C ABI, unit 2 of 4: the per-scale launches (chain / lattice / row kernels of
// wt_stencil.h, the bilateral march, run-time taps) and the decomposition drivers - the fused schedule, the
// overlapped multi-GPU passes, the pipelined host-to-host calls, the bilateral transform - plus wt_set_option.
// gfx950 only.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "wt_host.h"
#include "wt_kernels_common.h"
#include "wt_kernels_transform.h"
#include "wt_fused_decl.h"
#include "wt_unit_probe.h"

WT_UNIT_PROBE_DEFINE

// =============================================================================================
// run-time taps (user-defined scaling functions): the generic separable kernels
// =============================================================================================

// separable filter with the plan's run-time taps: rows into scratch 15, then columns (+ detail)
int launch_custom(wt_plan *p, const float *in, float *out_c, float *out_w, int s, int square, const char *name)
{
    if (s < 0 || s > 24) WT_FAIL("%s: scale %d out of range", name, s);
    float *tmp = nullptr;
    WT_TRY(plane_base(p, WT_PLANE_SCRATCH(15), &tmp));
    if (in == tmp || out_c == tmp || out_w == tmp) WT_FAIL("%s: scratch plane 15 is used internally for user-defined taps", name);
    if (out_c == in || out_w == in) WT_FAIL("%s: in-place operation", name);
    const CustomTaps t = plan_taps(p);
    const int d = 1 << s;
    dim3 grid((p->g.W + 255) / 256, (unsigned)std::min(p->g.nrows, 32768)), block(256);
    ProfScope ps(p->ctx, "wt_custom_kernels");
    hipLaunchKernelGGL(wt_custom_rows_kernel, grid, block, 0, p->ctx->stream, in, tmp, p->g, d, t, square);
    hipLaunchKernelGGL(wt_custom_cols_kernel, grid, block, 0, p->ctx->stream, (const float *)tmp, in, out_c, out_w, p->g, d, t);
    WT_HIP(hipGetLastError());
    return 0;
}

// sdev_loc (watroo/wavelets.py:24-32) with run-time taps: both moments through the generic
// separable kernels (mean in scratch 13), then the clip / sqrt / factors
int launch_custom_variance(wt_plan *p, const float *in, float *out, int s, float f1, float f2, int take_sqrt,
                                  const char *name)
{
    float *mean = nullptr;
    WT_TRY(plane_base(p, WT_PLANE_SCRATCH(13), &mean));
    if (in == mean || out == mean) WT_FAIL("%s: scratch plane 13 is used internally for user-defined taps", name);
    WT_TRY(launch_custom(p, in, mean, nullptr, s, 0, name));
    WT_TRY(launch_custom(p, in, out, nullptr, s, 1, name));
    const int64_t n4 = plan_n4(p);
    ProfScope ps(p->ctx, "wt_var_moments_kernel");
    hipLaunchKernelGGL(wt_var_moments_kernel, dim3(flat_grid(n4)), dim3(256), 0, p->ctx->stream, (const float *)mean,
                       (const float *)out, out, n4, f1, f2, take_sqrt);
    WT_HIP(hipGetLastError());
    return 0;
}

// =============================================================================================
// chain-march launches (generic per-scale operator)
// =============================================================================================
int check_scale(const wt_plan *p, int s, const char *who)
{
    if (s < 0 || s > 24) WT_FAIL("%s: scale %d out of range", who, s);
    const int hw = family_taps(p->family) / 2;
    const int64_t halo = (int64_t)hw << s;
    if (p->nranks > 1) {
        if (halo > p->g.halo) WT_FAIL("%s: scale %d needs %lld halo rows, plan has %d", who, s, (long long)halo, p->g.halo);
    }
    return 0;
}


// chunking of the polyphase chains (wt_stencil_launch.h)
static int chain_geometry(const wt_plan *p, int s, ChainArgs &a, dim3 &grid, dim3 &block, int gx_override = 0)
{
    return wt_chain_geometry<float>(p->g, s, a, grid, block, gx_override);
}

// tuning / A-B switches (wt_set_option)
// 0 forces the generic addressing of the fused passes (read by the launch code of every wt_fused_tu.hip unit)
int g_opt_fused_fast = getenv("WT_FUSED_NO_FAST") ? 0 : 1;
int g_opt_row_kernel = getenv("WT_NO_ROW_KERNEL") ? 0 : 1;      // (read by wt_stencil_launch.h in both units)
int g_opt_lattice = getenv("WT_NO_LATTICE") ? 0 : 1;
int g_opt_bilateral_paired = 1;   // bilateral march: one 8-byte load per operand pair (0: the generic two-load path)
// multi-GPU: run the halo exchange of pass i+1 beside the interior rows of pass i (0 = every
// exchange on the compute stream, between the passes)
int g_opt_overlap = getenv("WT_NO_OVERLAP") ? 0 : 1;
// compute units (of 256) the interior launch leaves free for the RCCL kernels of that exchange
int g_opt_overlap_reserve = getenv("WT_OVERLAP_RESERVE") ? atoi(getenv("WT_OVERLAP_RESERVE")) : 16;
// measurement aid: split the passes of a strip plan as the overlapped schedule does, without any
// exchange (FLAG_NO_EXCHANGE runs on one GPU: what do the edge / interior launches cost?)
int g_opt_split_dry = 0;

static void wt_set_hist_window(int on);
// wt_decompose_sum_host: pipeline the PCIe legs with the passes (0: upload, passes, download in turn)
int g_opt_host_pipeline = getenv("WT_NO_HOST_PIPELINE") ? 0 : 1;

extern "C" int wt_set_option(const char *name, int value)
{
    if (!name) WT_FAIL("wt_set_option: null name");
    if (!strcmp(name, "row_kernel")) { g_opt_row_kernel = value != 0; return 0; }
    if (!strcmp(name, "lattice_kernel")) { g_opt_lattice = value != 0; return 0; }
    if (!strcmp(name, "bilateral_paired")) { g_opt_bilateral_paired = value != 0; return 0; }
    if (!strcmp(name, "overlap")) { g_opt_overlap = value != 0; return 0; }
    if (!strcmp(name, "wow_overlap")) { g_opt_wow_overlap = value != 0; return 0; }
    if (!strcmp(name, "axis_filter")) { g_opt_axis_filter = value != 0; return 0; }
    if (!strcmp(name, "overlap_reserve")) { g_opt_overlap_reserve = value < 0 ? 0 : (value > 128 ? 128 : value); return 0; }
    if (!strcmp(name, "split_dry")) { g_opt_split_dry = value != 0; return 0; }
    if (!strcmp(name, "fused_fast")) { g_opt_fused_fast = value != 0; return 0; }
    if (!strcmp(name, "tri4")) { g_opt_tri4 = value != 0; return 0; }
    if (!strcmp(name, "host_pipeline")) { g_opt_host_pipeline = value != 0; return 0; }
    if (!strcmp(name, "fused64")) { wt_set_fused64(value != 0); return 0; }
    if (!strcmp(name, "select64_list")) { wt_set_select64_list(value != 0); return 0; }
    if (!strcmp(name, "f64_pairs")) { wt_set_f64_pairs(value != 0); return 0; }
    if (!strcmp(name, "stencil64")) { wt_set_stencil64(value != 0); return 0; }
    if (!strcmp(name, "hist_window")) { wt_set_hist_window(value != 0); return 0; }
    if (!strcmp(name, "scatter")) { g_opt_scatter = value < 0 ? 0 : (value > 16 ? 16 : value); return 0; }
    if (!strcmp(name, "scatter_strips")) { g_opt_scatter_strips = value != 0; return 0; }
    WT_FAIL("wt_set_option: unknown option '%s'", name);
}

template <int MODE>
static int launch_chain_args(wt_plan *p, ChainArgs a, int s, const char *name)
{
    if (p->ntaps) {      // user-defined scaling function: generic separable kernels
        if (MODE == MODE_SMOOTH || MODE == MODE_SMOOTH_SQ || MODE == MODE_DECOMP)
            return launch_custom(p, a.in, a.out_c, MODE == MODE_DECOMP ? a.out_w : nullptr, s, MODE == MODE_SMOOTH_SQ, name);
        if (MODE == MODE_VAR) return launch_custom_variance(p, a.in, a.out_c, s, a.f1, a.f2, a.take_sqrt, name);
        WT_FAIL("%s: not available with user-defined taps", name);
    }
    return wt32_stencil_launch(stencil_ctx(p), MODE, a, s, name);      // (wt_stencil32.hip)
}

template <int MODE>
static int launch_chain(wt_plan *p, const float *in, float *out_c, float *out_w, int s, float f1,
                        float f2, int take_sqrt, const char *name)
{
    ChainArgs a{};
    a.in = in; a.out_c = out_c; a.out_w = out_w; a.aux = nullptr;
    a.f1 = f1; a.f2 = f2; a.take_sqrt = take_sqrt;
    return launch_chain_args<MODE>(p, a, s, name);
}

int launch_chain_mode(wt_plan *p, int mode, ChainArgs a, int s, const char *name)
{
    switch (mode) {
        case MODE_SMOOTH: return launch_chain_args<MODE_SMOOTH>(p, a, s, name);
        case MODE_SMOOTH_SQ: return launch_chain_args<MODE_SMOOTH_SQ>(p, a, s, name);
        case MODE_DECOMP: return launch_chain_args<MODE_DECOMP>(p, a, s, name);
        case MODE_VAR: return launch_chain_args<MODE_VAR>(p, a, s, name);
        case MODE_WOW: return launch_chain_args<MODE_WOW>(p, a, s, name);
        case MODE_WOW_PLAIN: return launch_chain_args<MODE_WOW_PLAIN>(p, a, s, name);
        case MODE_WOW_GAMMA: return launch_chain_args<MODE_WOW_GAMMA>(p, a, s, name);
    }
    WT_FAIL("%s: unknown stencil mode %d", name, mode);
}

int maybe_exchange(wt_plan *p, int plane, int64_t rows, int flags)
{
    if (p->nranks == 1 || (flags & 2)) return 0;
    return wt_halo_exchange(p, plane, rows);
}


extern "C" int wt_atrous_scale(wt_plan *p, int src, int dst_c, int dst_w, int s, int flags)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_atrous_scale: null plan");
    WT_TRY(check_scale(p, s, "wt_atrous_scale"));
    if (src == dst_c || src == dst_w || dst_c == dst_w) WT_FAIL("wt_atrous_scale: planes must be distinct");
    float *in = nullptr, *oc = nullptr, *ow = nullptr;
    WT_TRY(plane_base(p, src, &in));
    WT_TRY(plane_base(p, dst_c, &oc));
    if (dst_w != WT_PLANE_NONE) WT_TRY(plane_base(p, dst_w, &ow));
    WT_TRY(maybe_exchange(p, src, scale_halo(p, s), flags));
    return launch_chain<MODE_DECOMP>(p, in, oc, ow, s, 1.f, 1.f, 0, "wt_chain_kernel<decomp>");
}

extern "C" int wt_smooth(wt_plan *p, int src, int dst, int s, int square_input, int flags)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_smooth: null plan");
    WT_TRY(check_scale(p, s, "wt_smooth"));
    if (src == dst) WT_FAIL("wt_smooth: src and dst must differ");
    float *in = nullptr, *o = nullptr;
    WT_TRY(plane_base(p, src, &in));
    WT_TRY(plane_base(p, dst, &o));
    WT_TRY(maybe_exchange(p, src, scale_halo(p, s), flags));
    if (square_input) return launch_chain<MODE_SMOOTH_SQ>(p, in, o, nullptr, s, 1.f, 1.f, 0, "wt_chain_kernel<smooth_sq>");
    return launch_chain<MODE_SMOOTH>(p, in, o, nullptr, s, 1.f, 1.f, 0, "wt_chain_kernel<smooth>");
}

extern "C" int wt_local_variance(wt_plan *p, int src, int dst, int s, float f1, float f2, int take_sqrt, int flags)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_local_variance: null plan");
    WT_TRY(check_scale(p, s, "wt_local_variance"));
    if (src == dst) WT_FAIL("wt_local_variance: src and dst must differ");
    float *in = nullptr, *o = nullptr;
    WT_TRY(plane_base(p, src, &in));
    WT_TRY(plane_base(p, dst, &o));
    WT_TRY(maybe_exchange(p, src, scale_halo(p, s), flags));
    return launch_chain<MODE_VAR>(p, in, o, nullptr, s, f1, f2, take_sqrt, "wt_chain_kernel<variance>");
}

// var == nullptr: the kernel forms the variance itself (times f1, f2) from its register window
// beside_side_work: the launch leaves room for the side stream's kernels (wt_decompose_bilateral with the overlap on)
static int launch_bilateral(wt_plan *p, const float *in, const float *var, float *out, float *out_w, int s,
                            float f1 = 1.f, float f2 = 1.f, int rev = 0, bool beside_side_work = false)
{
    if (p->g.border != 0 && p->g.border != 1) WT_FAIL("bilateral kernels implement the symmetric border (whole image or polyphase) only");
    if (p->ntaps) {      // user-defined scaling function: generic kernel, variance plane in scratch 12
        if (s < 0 || s > 24) WT_FAIL("wt_bilateral_conv: scale %d out of range", s);
        if (!var) {
            float *v = nullptr;
            WT_TRY(plane_base(p, WT_PLANE_SCRATCH(12), &v));
            if (in == v || out == v || out_w == v) WT_FAIL("wt_decompose_bilateral: scratch plane 12 is used internally for user-defined taps");
            WT_TRY(launch_custom_variance(p, in, v, s, f1, f2, 0, "wt_decompose_bilateral"));
            var = v;
        }
        dim3 grid((p->g.W + 255) / 256, (unsigned)std::min(p->g.nrows, 32768)), block(256);
        ProfScope ps(p->ctx, "wt_bilateral_custom_kernel");
        hipLaunchKernelGGL(wt_bilateral_custom_kernel, grid, block, 0, p->ctx->stream, in, var, out, out_w, p->g, p->g.H, 0,
                           1 << s, plan_taps(p), rev);
        WT_HIP(hipGetLastError());
        return 0;
    }
    ChainArgs a{};
    a.in = in; a.out_c = out; a.out_w = out_w; a.aux = var;
    a.inline_var = var == nullptr; a.f1 = f1; a.f2 = f2;
    dim3 grid, block;
    const bool b3 = p->family == WT_B3SPLINE;
    // 4 waves side by side on one chain item (the LDS ring holds 256 threads)
    WT_TRY(wt_march_geometry<float>(p->g, s, a, grid, block, ((p->g.W + 1) / 2 + 63) / 64, 4));
    // WT_PROF_SCALES=1: one profiler entry per dilation (tools/bench_bil.py)
    static const bool by_scale = getenv("WT_PROF_SCALES") != nullptr;
    static char names[32][40];
    if (s < 0 || s > 30) WT_FAIL("bilateral scale %d out of range", s);
    if (by_scale && !names[s][0]) snprintf(names[s], sizeof names[s], "wt_bilateral2_kernel d=%d", 1 << s);
    ProfScope ps(p->ctx, by_scale ? names[s] : "wt_bilateral2_kernel");
    // The march holds 20 KB of LDS and 4 x 96 VGPRs per workgroup: five fit a CU and take nearly every register of
    // it, so the wow updates that run on the side stream beside the NEXT scales (memory-bound, 98-142 VGPRs per wave)
    // find room only where a workgroup has just retired.  Beside side work the launch asks for 12 KB of dynamic LDS
    // it never touches: four workgroups per CU (the march loses nothing alone: 4 and 5 waves per SIMD time the same),
    // a quarter of the registers and 32 KB of LDS stay free.  cfg5 at 8192^2: 5.60-5.64 -> 5.45-5.46 ms
    // (profiles/r06_b_cfg5_lds_cap.txt).  WT_BIL_LDS_PAD overrides (experiments).
    static const int pad_env = getenv("WT_BIL_LDS_PAD") ? atoi(getenv("WT_BIL_LDS_PAD")) : -1;
    const int lds_pad = pad_env >= 0 ? pad_env : (beside_side_work && a.inline_var && b3 ? 12288 : 0);
    const bool paired = p->g.border == 0 && g_opt_bilateral_paired;      // one 8-byte load per operand pair (wt_kernels_transform.h)
#define WT_BIL2(KK, IV, PR) hipLaunchKernelGGL((wt_bilateral2_kernel<KK, IV, PR>), grid, block, lds_pad, p->ctx->stream, a)
    if (b3 && a.inline_var) { if (paired) WT_BIL2(5, true, true); else WT_BIL2(5, true, false); }
    else if (b3) { if (paired) WT_BIL2(5, false, true); else WT_BIL2(5, false, false); }
    else if (a.inline_var) { if (paired) WT_BIL2(3, true, true); else WT_BIL2(3, true, false); }
    else { if (paired) WT_BIL2(3, false, true); else WT_BIL2(3, false, false); }
#undef WT_BIL2
    WT_HIP(hipGetLastError());
    return 0;
}

extern "C" int wt_bilateral_conv(wt_plan *p, int src, int var, int dst, int s, int flags)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_bilateral_conv: null plan");
    WT_TRY(check_scale(p, s, "wt_bilateral_conv"));
    if (src == dst || var == dst) WT_FAIL("wt_bilateral_conv: dst must differ from src and var");
    float *in = nullptr, *v = nullptr, *o = nullptr;
    WT_TRY(plane_base(p, src, &in));
    WT_TRY(plane_base(p, var, &v));
    WT_TRY(plane_base(p, dst, &o));
    WT_TRY(maybe_exchange(p, src, scale_halo(p, s), flags));
    return launch_bilateral(p, in, v, o, nullptr, s, 1.f, 1.f, (flags & 8) != 0);
}

// =============================================================================================
// decomposition drivers
// =============================================================================================
// One pass of the schedule: scales [s0, s0+ns) from plane `cur` (= c_{s0}) into the detail
// planes s0..s0+ns-1 and plane `nxt` (= c_{s0+ns}).
// acc / p_sum: 0 = plain pass; 1 / 2 = the pass also carries the plane sum in plane `p_sum`
// (2 = last pass of the schedule: the smooth plane is added too) - fused passes only.
// flag bit4: the fused first pass of a plain decomposition also histograms |w_0| (first radix level
// of wt_abs_median's select).  begin: clear the bins once per entry point (a pass may be several
// launches); end: leave the marker if the histogram variant really ran.
// (src: the plane the first pass reads.  Whole images of at least 2^20 pixels with a built-in family get
// the WINDOWED histogram: wt_median_window_kernel predicts where the median of |w_0| lies from 4096
// pixels of `src`, and the first pass bins 21-bit keys around it - wt_abs_median then needs ONE more
// pass over the plane instead of two.  wt_set_option("hist_window", 0) keeps the plain 11-bit bins.)
int g_opt_hist_window = getenv("WT_NO_HIST_WINDOW") ? 0 : 1;
static void wt_set_hist_window(int on) { g_opt_hist_window = on; }
static int prehist_begin(wt_plan *p, int flags, int src = WT_PLANE_NONE)
{
    wt_ctx *c = p->ctx;
    c->prehist_ran = false;
    if (flags & 16) {
        c->prehist_plan = nullptr;                       // the bins are about to be cleared
        c->prehist_windowed = false;
        WT_HIP(hipMemsetAsync(c->d_hist, 0, WT_HIST_BINS * sizeof(uint32_t), c->stream));
        if (g_opt_hist_window && src != WT_PLANE_NONE && p->nranks == 1 && !p->g.border && !p->ntaps &&
            (int64_t)p->g.H * p->g.W >= ((int64_t)1 << 20) && p->g.H >= 64 && p->g.W >= 64) {
            float *in = nullptr;
            WT_TRY(plane_base(p, src, &in));
            ProfScope ps(c, "wt_median_window_kernel");
            uint32_t *keys = (uint32_t *)c->d_partials;      // 16 KB of the reduction scratch (stream-ordered use)
            if (p->family == WT_B3SPLINE) hipLaunchKernelGGL((wt_median_sample_kernel<5, float>), dim3(64), dim3(64), 0, c->stream, (const float *)in, p->g, keys);
            else hipLaunchKernelGGL((wt_median_sample_kernel<3, float>), dim3(64), dim3(64), 0, c->stream, (const float *)in, p->g, keys);
            hipLaunchKernelGGL(wt_median_window_kernel, dim3(1), dim3(1024), 0, c->stream, (const uint32_t *)keys, hist_base_word(c));
            WT_HIP(hipGetLastError());
            c->prehist_windowed = true;
        }
    }
    return 0;
}
static void prehist_end(wt_plan *p)
{
    if (p->ctx->prehist_ran) {
        p->ctx->prehist_plan = p;
        p->ctx->prehist_plane = 0;
    }
    p->ctx->prehist_ran = false;
}

static int decompose_pass_impl(wt_plan *p, int cur, int nxt, int s0, int ns, int flags, int acc, bool first_of_sum,
                               int p_sum, const FusedRows &rows = FusedRows())
{
    if (!p) WT_FAIL("wt_decompose_pass: null plan");
    if (ns < 1 || ns > WT_FUSED_MAX_SCALES || s0 < 0 || s0 + ns - 1 > p->max_level)
        WT_FAIL("wt_decompose_pass: scales [%d,%d) outside the plan (max_level %d)", s0, s0 + ns, p->max_level);
    if (cur == nxt || (cur >= s0 && cur < s0 + ns) || (nxt >= s0 && nxt < s0 + ns))
        WT_FAIL("wt_decompose_pass: input/output planes alias the detail planes of the pass");
    const int hw = family_taps(p->family) / 2;
    const int halo = hw * ((1 << (s0 + ns)) - (1 << s0));
    if (p->nranks > 1 && halo > p->g.halo) WT_FAIL("wt_decompose_pass: pass needs %d halo rows, plan has %d", halo, p->g.halo);
    WT_TRY(maybe_exchange(p, cur, halo, flags));
    float *in = nullptr, *oc = nullptr;
    WT_TRY(plane_base(p, cur, &in));
    WT_TRY(plane_base(p, nxt, &oc));
    // a single scale: the per-scale kernels - except the scale that ends a 4- or 7-scale fused
    // schedule, which has a fused kernel of its own (same bits with and without the carried sum)
    if (ns == 1 && !((flags & 1) && !p->g.border && !p->ntaps && wt_fused_supported(p) && wt_fused_has_pass(s0, 1, p->family))) {
        if (acc) WT_FAIL("wt_decompose_pass_sum: no accumulate kernel for the single scale %d", s0);
        if (rows.n) WT_FAIL("wt_decompose_pass: row ranges need a fused pass");
        WT_TRY(check_scale(p, s0, "wt_decompose_pass"));
        float *ow = nullptr;
        WT_TRY(plane_base(p, s0, &ow));
        return launch_chain<MODE_DECOMP>(p, in, oc, ow, s0, 1.f, 1.f, 0, "wt_chain_kernel<decomp>");
    }
    if (p->g.border) WT_FAIL("wt_decompose_pass: fused passes implement the symmetric border only (use flags without bit0)");
    if (!wt_fused_has_pass(s0, ns, p->family)) WT_FAIL("wt_decompose_pass: no fused kernel for first scale %d x %d scales", s0, ns);
    float *ow[WT_FUSED_MAX_SCALES] = {nullptr};
    for (int k = 0; k < ns; ++k) WT_TRY(plane_base(p, s0 + k, &ow[k]));
    float *ps = nullptr;
    if (acc && first_of_sum != (s0 == 0))
        WT_FAIL("wt_decompose_pass_sum: first must be set for the pass that starts at scale 0 and only for it (got first=%d, s0=%d)", (int)first_of_sum, s0);
    if (acc) WT_TRY(plane_base(p, p_sum, &ps));
    if (acc == 0 && (flags & 16) && s0 == 0) {
        // plain first pass that also histograms the first radix level of |w_0| for wt_abs_median
        // (the caller cleared the bins: a pass may be several launches)
        p->ctx->prehist_ran = true;
        return wt_fused_launch(p, in, oc, ow, s0, ns, 3, nullptr, nullptr, rows, p->ctx->d_hist,
                               p->ctx->prehist_windowed ? hist_base_word(p->ctx) : nullptr);
    }
    return wt_fused_launch(p, in, oc, ow, s0, ns, acc, first_of_sum ? nullptr : ps, ps, rows);
}

extern "C" int wt_decompose_pass(wt_plan *p, int cur, int nxt, int s0, int ns, int flags)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_decompose_pass: null plan");
    WT_TRY(prehist_begin(p, flags, s0 == 0 ? cur : WT_PLANE_NONE));
    WT_TRY(decompose_pass_impl(p, cur, nxt, s0, ns, flags, 0, false, WT_PLANE_NONE));
    prehist_end(p);
    return 0;
}

extern "C" int wt_decompose_pass_sum(wt_plan *p, int cur, int nxt, int s0, int ns, int flags, int sum_plane, int first,
                                     int last)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_decompose_pass_sum: null plan");
    if (!wt_fused_has_pass(s0, ns, p->family)) WT_FAIL("wt_decompose_pass_sum: no fused kernel for first scale %d x %d scales", s0, ns);
    if (sum_plane == cur || sum_plane == nxt || (sum_plane >= s0 && sum_plane < s0 + ns))
        WT_FAIL("wt_decompose_pass_sum: the sum plane aliases a plane of the pass");
    return decompose_pass_impl(p, cur, nxt, s0, ns, flags, last ? 2 : 1, first != 0, sum_plane);
}

// The passes of a schedule.  Multi-GPU strips with the overlap option: every exchange runs on the
// communication stream, and a fused pass whose output plane the NEXT pass needs halos of is split
// into its edge rows (the rows the neighbours need: launched first), the exchange of exactly
// those rows (communication stream, after the edge launch) and its interior rows (compute
// stream, beside the exchange; the grid leaves a few workgroup slots to the RCCL kernels).  Same
// kernels, same per-pixel arithmetic: bit-identical to the serial order.
static int run_schedule(wt_plan *p, int src, int level, int flags, const int32_t *tr, int np, bool with_sum, int dst)
{
    wt_ctx *c = p->ctx;
    const bool multi = p->nranks > 1 && !(flags & 2);
    const bool dry = g_opt_split_dry && p->nranks > 1 && (flags & 2);
    const bool overlap = dry || (multi && g_opt_overlap && c->comm_stream);
    auto exchange_async = [&](int plane, int64_t rows, int pass) -> int {     // after everything queued on the compute stream so far
        if (dry) return 0;
        WT_HIP(hipEventRecord(c->ev_to_comm, c->stream));
        WT_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_to_comm, 0));
        char nm[40];
        snprintf(nm, sizeof nm, "rccl_halo_exchange/pass%d", pass);
        WT_TRY(halo_exchange_on(p, plane, rows, c->comm_stream, nm));
        WT_HIP(hipEventRecord(c->ev_from_comm, c->comm_stream));
        return 0;
    };
    // Overlapped schedule (round 3: every exchange hides, the first one included).  A pass needs its
    // neighbours' rows only for the output rows within `halo` of a strip boundary.  So for every pass:
    //   1. the exchange of the pass INPUT's halo rows starts on the communication stream (behind
    //      everything queued so far, i.e. behind the previous pass),
    //   2. the INTERIOR rows [halo, nrows - halo) - which read own rows only - are launched at once and
    //      run beside the exchange (the launch leaves `overlap_reserve` CUs to the RCCL kernels),
    //   3. the edge rows follow when the exchange has landed.
    // Until round 2 a pass ran its edge rows FIRST and the next pass's exchange beside its interior,
    // which left the exchange of the very first pass (the input image's 14 rows) with nothing to hide
    // behind.  Same launches on the same row ranges, so the bits do not change.
    int cur = src;
    for (int i = 0; i < np; ++i) {
        const int s0 = tr[3 * i], ns = tr[3 * i + 1];
        const bool last = s0 + ns == level;
        const int nxt = last ? level : WT_PLANE_SCRATCH(i & 1);
        const int acc = with_sum ? (last ? 2 : 1) : 0;
        if (!overlap) {
            WT_TRY(decompose_pass_impl(p, cur, nxt, s0, ns, flags, acc, i == 0, dst));
            cur = nxt;
            continue;
        }
        const int64_t halo = tr[3 * i + 2];
        const int nrows = p->g.nrows;
        const bool up = p->rank > 0, dn = p->rank + 1 < p->nranks;
        // (a pass of one scale runs a fused kernel - and can take row ranges - only where one is built)
        const bool ranged = ns > 1 || ((flags & 1) && !p->g.border && !p->ntaps && wt_fused_supported(p) && wt_fused_has_pass(s0, 1, p->family));
        if (halo > 0 && ranged && (up || dn) && 2 * halo < nrows) {
            WT_TRY(exchange_async(cur, halo, i));
            FusedRows edge, inner;
            edge.part = 2;
            inner.part = 1;
            if (up) { edge.lo[edge.n] = 0; edge.hi[edge.n] = (int)halo; edge.n++; }
            if (dn) { edge.lo[edge.n] = nrows - (int)halo; edge.hi[edge.n] = nrows; edge.n++; }
            inner.n = 1;
            inner.lo[0] = up ? (int)halo : 0;
            inner.hi[0] = dn ? nrows - (int)halo : nrows;
            inner.reserve = g_opt_overlap_reserve;
            WT_TRY(decompose_pass_impl(p, cur, nxt, s0, ns, flags | 2, acc, i == 0, dst, inner));
            if (!dry) WT_HIP(hipStreamWaitEvent(c->stream, c->ev_from_comm, 0));
            WT_TRY(decompose_pass_impl(p, cur, nxt, s0, ns, flags | 2, acc, i == 0, dst, edge));
        } else {
            if (halo > 0) {
                WT_TRY(exchange_async(cur, halo, i));
                if (!dry) WT_HIP(hipStreamWaitEvent(c->stream, c->ev_from_comm, 0));
            }
            WT_TRY(decompose_pass_impl(p, cur, nxt, s0, ns, flags | 2, acc, i == 0, dst));
        }
        cur = nxt;
    }
    return 0;
}

extern "C" int wt_decompose_sum(wt_plan *p, int src, int level, int dst, int flags)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_decompose_sum: null plan");
    if (level < 0 || level > p->max_level) WT_FAIL("wt_decompose_sum: level %d exceeds plan max_level %d", level, p->max_level);
    if (src >= 0 && src <= level) WT_FAIL("wt_decompose_sum: src plane %d is one of the output planes", src);
    if (dst >= 0 && dst <= level) WT_FAIL("wt_decompose_sum: dst plane %d is one of the output planes", dst);
    if (dst == src) WT_FAIL("wt_decompose_sum: dst and src must differ");
    if (src == WT_PLANE_SCRATCH(0) || src == WT_PLANE_SCRATCH(1) || dst == WT_PLANE_SCRATCH(0) || dst == WT_PLANE_SCRATCH(1))
        WT_FAIL("wt_decompose_sum: scratch planes 0/1 are used internally");
    int32_t tr[3 * 32];
    int np = 0;
    bool fusable = (flags & 1) && level > 0 && !p->g.border && !p->ntaps && wt_fused_supported(p);
    if (fusable) {
        WT_TRY(wt_schedule(p->family, level, 1, tr, 32, &np));
        for (int i = 0; i < np; ++i) fusable = fusable && wt_fused_has_pass(tr[3 * i], tr[3 * i + 1], p->family);
    }
    if (!fusable) {      // a schedule with single-scale passes: the two-step form
        WT_TRY(wt_decompose(p, src, level, flags));
        return wt_plane_sum(p, 0, level + 1, dst);
    }
    return run_schedule(p, src, level, flags, tr, np, true, dst);
}

// Would wt_decompose_sum(plan, ., level, ., FLAG_FUSED) run as accumulate passes (every pass of the
// schedule has a fused kernel, symmetric border, built-in taps, rows short enough)?  Host logic.
// Host-to-host form of wt_decompose_sum, pipelined over PCIe (round 3).
//   serial:     upload 4.7 ms | passes 0.7 ms | download 4.7 ms      (8192^2, 57 GB/s per direction)
//   pipelined:  the image goes up in blocks of rows on a transfer stream; as soon as the rows a pass
//               needs (its own rows + the pass's halo) are there the pass runs on them (row
//               sub-ranges of the fused kernels: same per-pixel arithmetic, identical bits); the
//               rows of the reconstruction that the last pass has finished go down on a second
//               transfer stream while later blocks are still coming up - PCIe is full duplex, so
//               the call costs about one leg plus one block of latency instead of two legs.
// Device state afterwards is that of the serial sequence: PLANE_INPUT holds the image, planes
// 0..level the coefficients, `dst` the reconstruction.  Planes mapped over scattered chunks cannot
// be the target of a 2-D memcpy: blocks bounce through the plan's contiguous stage plane (a copy
// kernel per block, hidden behind the transfers); the stage rows of a block are reused for the
// reconstruction rows once the block has been copied on (stream order).
// Threshold step of the pipelined host call (wt_denoise_sum_host): Coefficients.denoise over the first
// n_den planes - the planes of the first k_passes passes of the schedule - fused with the start of the
// plane sum, between those passes and the ones that carry the sum on.
struct HostDenoise {
    int k_passes, n_den, soft;
    const double *tau, *wgt;
};

static int host_pipeline(wt_plan *p, const float *host_in, int64_t in_stride, int level, int dst, float *host_out, int64_t out_stride,
                         int block_rows, const HostDenoise *den)
{
    if (!p || !host_in || !host_out) WT_FAIL("wt_decompose_sum_host: null pointer");
    if (in_stride < p->g.W || out_stride < p->g.W) WT_FAIL("wt_decompose_sum_host: host stride below the width %d", p->g.W);
    if (level < 0 || level > p->max_level) WT_FAIL("wt_decompose_sum_host: level %d exceeds plan max_level %d", level, p->max_level);
    if ((dst >= 0 && dst <= level) || dst == WT_PLANE_INPUT || dst == WT_PLANE_SCRATCH(0) || dst == WT_PLANE_SCRATCH(1))
        WT_FAIL("wt_decompose_sum_host: dst plane %d is an input / output / internal plane of the transform", dst);
    wt_ctx *c = p->ctx;
    const int H = p->g.nrows, P = p->g.P, W = p->g.W;
    int32_t tr[3 * 32];
    int np = 0;
    bool pipe = g_opt_host_pipeline && p->nranks == 1 && p->g.row0 == 0 && p->g.nrows == p->g.H && level > 0 && !p->g.border && !p->ntaps &&
                wt_fused_supported(p);
    if (pipe) {
        WT_TRY(wt_schedule(p->family, level, 1, tr, 32, &np));
        for (int i = 0; i < np; ++i) pipe = pipe && wt_fused_has_pass(tr[3 * i], tr[3 * i + 1], p->family);
    }
    if (block_rows <= 0) block_rows = std::max(256, (H + 15) / 16);  // sixteen blocks: measured best at 8192^2 (tail = one block of each leg)
    block_rows = (block_rows + 63) / 64 * 64;
    if (pipe && (H < 2 * block_rows || (int64_t)H * W < (1 << 22))) pipe = false;      // small images: nothing to overlap
    if (den) {
        // the threshold step sits between two passes of an all-fused schedule and covers exactly the
        // planes of the passes before it; anything else is the caller's job (serial sequence)
        int covered = 0;
        for (int i = 0; i < np && i < den->k_passes; ++i) covered += tr[3 * i + 1];
        if (!pipe || den->k_passes < 1 || den->k_passes >= np || covered != den->n_den)
            WT_FAIL("wt_denoise_sum_host: the threshold step must follow the first k passes (0 < k < passes) of a fused schedule and cover "
                    "their planes (got k = %d, n_den = %d, %d passes%s)", den->k_passes, den->n_den, np, pipe ? "" : ", no pipeline for this plan / size");
    }
    if (!pipe) {
        WT_TRY(wt_upload(p, WT_PLANE_INPUT, host_in, in_stride));
        WT_TRY(wt_decompose_sum(p, WT_PLANE_INPUT, level, dst, 1));
        return wt_download(p, dst, host_out, out_stride);
    }
    WT_HIP(hipSetDevice(c->device));
    if (!c->xfer_in) {
        WT_HIP(hipStreamCreateWithFlags(&c->xfer_in, hipStreamNonBlocking));
        WT_HIP(hipStreamCreateWithFlags(&c->xfer_out, hipStreamNonBlocking));
    }
    float *in_b = nullptr, *out_b = nullptr, *stage = nullptr;
    WT_TRY(plane_base(p, WT_PLANE_INPUT, &in_b));
    WT_TRY(plane_base(p, dst, &out_b));
    for (int s = 0; s <= level; ++s) {                       // (allocate before the first launch; drops the median marker)
        float *t = nullptr;
        WT_TRY(plane_base(p, s, &t));
    }
    const bool vin = is_vmm(p, in_b), vout = is_vmm(p, out_b);
    if (vin || vout) WT_TRY(vmm_stage(p, &stage));
    float *up_b = vin ? stage : in_b, *down_b = vout ? stage : out_b;
    const size_t in_span = ((size_t)(H - 1) * (size_t)in_stride + (size_t)W) * 4, out_span = ((size_t)(H - 1) * (size_t)out_stride + (size_t)W) * 4;
    const bool pin_in = try_pin(host_in, in_span), pin_out = try_pin(host_out, out_span);
    std::vector<hipEvent_t> evs;
    auto new_event = [&](hipEvent_t *e) -> hipError_t {
        hipError_t rc = hipEventCreateWithFlags(e, hipEventDisableTiming);
        if (rc == hipSuccess) evs.push_back(*e);
        return rc;
    };
    std::vector<int> done(np, 0);
    int out_done = 0, rc = 0, den_done = 0;
    const int kd = den ? den->k_passes : 0;              // passes [0, kd) are plain, the threshold step follows them
    hipError_t e = hipSuccess;
    auto run = [&]() -> int {
        // the transfer streams start behind whatever the compute stream was doing to these planes
        hipEvent_t e0;
        WT_HIP(new_event(&e0));
        WT_HIP(hipEventRecord(e0, c->stream));
        WT_HIP(hipStreamWaitEvent(c->xfer_in, e0, 0));
        WT_HIP(hipStreamWaitEvent(c->xfer_out, e0, 0));
        // block boundaries: equal blocks, the last one cut into 1/2 + 1/4 + 1/4 (the tail of the call
        // is the passes and the download of whatever came up last)
        std::vector<int> cuts;
        for (int y = 0; y < H; y += block_rows) cuts.push_back(y);
        if (cuts.size() > 1 && H - cuts.back() > 192) {
            const int yl = cuts.back(), n = H - yl, q = (n / 4 + 63) / 64 * 64;
            if (n - 2 * q >= 64) {
                cuts.push_back(yl + n - 2 * q);
                cuts.push_back(H - q);
            }
        }
        cuts.push_back(H);
        // (rows that are contiguous on both sides go as ONE linear copy - the DMA engines' fast path -
        //  when the piece is large: below ~16 MiB the linear path is the slow one, measured)
        const size_t linear_min = (size_t)16 << 20;
        for (size_t bi = 0; bi + 1 < cuts.size(); ++bi) {
            const int y0 = cuts[bi], y1 = cuts[bi + 1];
            if (y1 <= y0) continue;
            if (in_stride == W && P == W && (size_t)(y1 - y0) * W * 4 >= linear_min)
                WT_HIP(hipMemcpyAsync(up_b + (size_t)y0 * P, host_in + (size_t)y0 * in_stride, (size_t)(y1 - y0) * W * 4, hipMemcpyHostToDevice, c->xfer_in));
            else
                WT_HIP(hipMemcpy2DAsync(up_b + (size_t)y0 * P, (size_t)P * 4, host_in + (size_t)y0 * in_stride, (size_t)in_stride * 4, (size_t)W * 4,
                                        (size_t)(y1 - y0), hipMemcpyHostToDevice, c->xfer_in));
            hipEvent_t eu;
            WT_HIP(new_event(&eu));
            WT_HIP(hipEventRecord(eu, c->xfer_in));
            WT_HIP(hipStreamWaitEvent(c->stream, eu, 0));
            if (vin) WT_TRY(copy2d(p, p, in_b + (size_t)y0 * P, (size_t)P, stage + (size_t)y0 * P, (size_t)P, (size_t)P, (size_t)(y1 - y0), c->stream));
            int avail = y1, cur = WT_PLANE_INPUT;
            for (int i = 0; i < np; ++i) {
                const int s0 = tr[3 * i], ns = tr[3 * i + 1], halo = tr[3 * i + 2];
                const bool last = s0 + ns == level;
                const int nxt = last ? level : WT_PLANE_SCRATCH(i & 1);
                if (den && i == kd && avail > den_done) {
                    // rows the plain passes have finished: thresholds + start of the sum (the sum-carrying
                    // passes below read these rows of `dst` only where they store, no halo)
                    WT_TRY(denoise_sum_rows(p, den->n_den, dst, den->n_den, den->tau, den->wgt, den->soft, den_done, avail));
                    den_done = avail;
                }
                const int ready = avail == H ? H : std::max(done[i], avail - halo);
                if (ready > done[i]) {
                    FusedRows rows;
                    rows.n = 1;
                    rows.lo[0] = done[i];
                    rows.hi[0] = ready;
                    const int acc = (den && i < kd) ? 0 : (last ? 2 : 1);
                    WT_TRY(decompose_pass_impl(p, cur, nxt, s0, ns, 1 | 2, acc, acc != 0 && i == 0, dst, rows));
                    done[i] = ready;
                }
                avail = done[i];
                cur = nxt;
            }
            const int fin = done[np - 1];
            if (fin > out_done) {
                if (vout) WT_TRY(copy2d(p, p, stage + (size_t)out_done * P, (size_t)P, out_b + (size_t)out_done * P, (size_t)P, (size_t)P,
                                        (size_t)(fin - out_done), c->stream));
                hipEvent_t ec;
                WT_HIP(new_event(&ec));
                WT_HIP(hipEventRecord(ec, c->stream));
                WT_HIP(hipStreamWaitEvent(c->xfer_out, ec, 0));
                if (out_stride == W && P == W && (size_t)(fin - out_done) * W * 4 >= linear_min)
                    WT_HIP(hipMemcpyAsync(host_out + (size_t)out_done * out_stride, down_b + (size_t)out_done * P, (size_t)(fin - out_done) * W * 4,
                                          hipMemcpyDeviceToHost, c->xfer_out));
                else
                    WT_HIP(hipMemcpy2DAsync(host_out + (size_t)out_done * out_stride, (size_t)out_stride * 4, down_b + (size_t)out_done * P, (size_t)P * 4,
                                            (size_t)W * 4, (size_t)(fin - out_done), hipMemcpyDeviceToHost, c->xfer_out));
                out_done = fin;
            }
        }
        return 0;
    };
    rc = run();
    // drain everything whatever happened (host buffers are unpinned below, events destroyed)
    hipError_t e1 = hipStreamSynchronize(c->xfer_in), e2 = hipStreamSynchronize(c->stream), e3 = hipStreamSynchronize(c->xfer_out);
    e = e1 != hipSuccess ? e1 : (e2 != hipSuccess ? e2 : e3);
    for (auto ev : evs) (void)hipEventDestroy(ev);
    if (pin_in) (void)hipHostUnregister(const_cast<float *>(host_in));
    if (pin_out) (void)hipHostUnregister(host_out);
    if (rc) return rc;
    WT_HIP(e);
    if (out_done != H) WT_FAIL("wt_decompose_sum_host: internal error, %d of %d rows delivered", out_done, H);
    return 0;
}

extern "C" int wt_decompose_sum_host(wt_plan *p, const float *host_in, int64_t in_stride, int level, int dst, float *host_out,
                                     int64_t out_stride, int block_rows)
{
    WtGuard guard_(ctx_of(p));
    return host_pipeline(p, host_in, in_stride, level, dst, host_out, out_stride, block_rows, nullptr);
}

// utils.denoise with the noise level GIVEN (watroo/utils.py:83-102 with noise=...: every threshold is
// known before the first pixel arrives), host to host: the image goes up in blocks of rows, the first
// k_passes passes of the fused schedule run on a block as its rows arrive, Coefficients.denoise over
// their n_den planes starts the plane sum (wt_denoise_sum on the finished rows; the planes are left
// as they are: denoise() does not return them), the remaining passes carry the sum, and finished
// rows of the result go down while later blocks are still coming up - about one PCIe leg instead of
// two.  Same kernels on row sub-ranges: the result equals upload + passes + wt_denoise_sum + passes +
// download bit for bit.  Needs 0 < k_passes < passes of an all-fused schedule and a size worth
// pipelining; otherwise an error (the caller runs the serial sequence).
extern "C" int wt_denoise_sum_host(wt_plan *p, const float *host_in, int64_t in_stride, int level, int k_passes, int n_den,
                                   const double *tau, const double *wgt, int soft, int dst, float *host_out, int64_t out_stride,
                                   int block_rows)
{
    WtGuard guard_(ctx_of(p));
    if (!tau || !wgt) WT_FAIL("wt_denoise_sum_host: null tau / wgt");
    if (n_den < 1 || n_den > WT_MAX_SUM_PLANES) WT_FAIL("wt_denoise_sum_host: n_den %d out of range", n_den);
    HostDenoise den{k_passes, n_den, soft, tau, wgt};
    return host_pipeline(p, host_in, in_stride, level, dst, host_out, out_stride, block_rows, &den);
}

extern "C" int wt_plan_fused_ok(wt_plan *p, int level, int *ok)
{
    WtGuard guard_(ctx_of(p));
    if (!p || !ok) WT_FAIL("wt_plan_fused_ok: null pointer");
    *ok = 0;
    if (level <= 0 || level > p->max_level || p->g.border || p->ntaps || !wt_fused_supported(p)) return 0;
    int32_t tr[3 * 32];
    int np = 0;
    WT_TRY(wt_schedule(p->family, level, 1, tr, 32, &np));
    for (int i = 0; i < np; ++i)
        if (!wt_fused_has_pass(tr[3 * i], tr[3 * i + 1], p->family)) return 0;
    *ok = 1;
    return 0;
}

extern "C" int wt_decompose(wt_plan *p, int src, int level, int flags)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_decompose: null plan");
    if (level < 0 || level > p->max_level) WT_FAIL("wt_decompose: level %d exceeds plan max_level %d", level, p->max_level);
    if (src >= 0 && src <= level) WT_FAIL("wt_decompose: src plane %d is one of the output planes", src);
    if (src == WT_PLANE_SCRATCH(0) || src == WT_PLANE_SCRATCH(1)) WT_FAIL("wt_decompose: scratch planes 0/1 are used internally");
    if (level == 0) return wt_copy_plane(p, src, 0);
    int32_t tr[3 * 32];
    int np = 0;
    if (p->ntaps) flags &= ~1;          // user-defined taps: one generic pass per scale
    WT_TRY(wt_schedule(p->family, level, (flags & 1) && wt_fused_supported(p), tr, 32, &np));
    WT_TRY(prehist_begin(p, flags, src));
    WT_TRY(run_schedule(p, src, level, flags, tr, np, false, WT_PLANE_NONE));
    prehist_end(p);
    return 0;
}

extern "C" int wt_decompose_bilateral(wt_plan *p, int src, int level, const double *sigma_b, int bilateral_scaling, int flags)
{
    WtGuard guard_(ctx_of(p));
    if (!p || !sigma_b) WT_FAIL("wt_decompose_bilateral: null pointer");
    if (level < 0 || level > p->max_level) WT_FAIL("wt_decompose_bilateral: level %d exceeds plan max_level %d", level, p->max_level);
    if (src >= 0 && src <= level) WT_FAIL("wt_decompose_bilateral: src plane %d is one of the output planes", src);
    if (src <= WT_PLANE_SCRATCH(0) && src >= WT_PLANE_SCRATCH(2)) WT_FAIL("wt_decompose_bilateral: scratch planes 0..2 are used internally");
    if (level == 0) return wt_copy_plane(p, src, 0);
    const bool overlap = g_opt_wow_overlap && p->nranks == 1;
    if (overlap) WT_TRY(wt_scale_events(p->ctx, p->scale_ev, level));
    int cur = src;
    for (int s = 0; s < level; ++s) {
        WT_TRY(check_scale(p, s, "wt_decompose_bilateral"));
        const int nxt = (s == level - 1) ? level : WT_PLANE_SCRATCH(s & 1);
        float *in = nullptr, *oc = nullptr, *ow = nullptr, *var = nullptr;
        WT_TRY(plane_base(p, cur, &in));
        WT_TRY(plane_base(p, nxt, &oc));
        WT_TRY(plane_base(p, s, &ow));
        WT_TRY(plane_base(p, WT_PLANE_SCRATCH(2), &var));
        WT_TRY(maybe_exchange(p, cur, scale_halo(p, s), flags));
        // variance = sdev_loc(c_s)^2-form * sigma_b[s]**2 (* (s+1))   watroo/wavelets.py:434-436
        const float f1 = (float)(sigma_b[s] * sigma_b[s]);
        const float f2 = bilateral_scaling ? (float)(s + 1) : 1.f;
        if (flags & 4) {   // two-kernel form (variance plane materialised), kept for A/B tests
            WT_TRY(launch_chain<MODE_VAR>(p, in, var, nullptr, s, f1, f2, 0, "wt_chain_kernel<variance>"));
            WT_TRY(launch_bilateral(p, in, var, oc, ow, s, 1.f, 1.f, (flags & 8) != 0));
        } else {
            WT_TRY(launch_bilateral(p, in, nullptr, oc, ow, s, f1, f2, (flags & 8) != 0, overlap));
        }
        if (overlap) WT_HIP(hipEventRecord(p->scale_ev[s], p->ctx->stream));     // w_s is written
        cur = nxt;
    }
    // the per-scale work on w_s that follows (wt_wow_scale, wt_abs_median) may run beside the scales still queued
    p->overlap_scales = overlap ? level : 0;
    p->overlap_ok = overlap;
    return 0;
}

