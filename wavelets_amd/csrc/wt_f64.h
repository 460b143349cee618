// float64 engine, host side: wt_plan64 and the wt64_* entry points of include/watroo_hip.h.
//
// The reference keeps float64 inputs in float64 and promotes int / big-endian inputs to float64
// (watroo/wavelets.py:297,319-320; the README examples are float64).  Images with a built-in family run on the
// tuned kernels - the fused double passes (wt_fused.h), the per-scale kernels for double (wt_stencil.h through
// wt_stencil64.hip), the float64 bilateral march (wt_bilateral64.h); signals, cubes and user-defined taps on the
// generic kernels of wt_kernels_f64.h; the exact median on wt_select64.h.
// The body of wt_f64.hip (its own translation unit since round 5).
#pragma once
#include "wt_kernels_f64.h"
#include "wt_select64.h"

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int plan64_base(wt_plan64 *p, int id, double **base)
{
    double **slot = nullptr;
    if (id >= 0 && id <= p->max_level) slot = &p->coef[id];
    else if (id == WT_PLANE_INPUT) slot = &p->input;
    else if (id == WT_PLANE_OUT) slot = &p->out;
    else if (id <= WT_PLANE_SCRATCH(0) && id > WT_PLANE_SCRATCH(WT64_NUM_SCRATCH)) slot = &p->scratch[-3 - id];
    else WT_FAIL("float64 plan: invalid plane id %d (max_level %d)", id, p->max_level);
    if (p->ctx->prehist_plan == p && p->ctx->prehist_plane == id) p->ctx->prehist_plan = nullptr;   // plane touched
    if (!p->ctx->in_side) {            // a main-stream access: behind everything the side stream has queued
        WT_TRY(wt_side_join(p->ctx));
        p->overlap_ok = false;
    }
    if (!*slot) {
        void *q = nullptr;
        WT_HIP(hipSetDevice(p->ctx->device));
        WT_HIP(hipMalloc(&q, (size_t)p->g.nrows * p->g.P * sizeof(double)));
        p->allocs.push_back(q);
        *slot = (double *)q;
    }
    *base = *slot;
    return 0;
}

static int plan64_tmp(wt_plan64 *p, int i, double **base)
{
    if (!p->tmp[i]) {
        void *q = nullptr;
        WT_HIP(hipSetDevice(p->ctx->device));
        WT_HIP(hipMalloc(&q, (size_t)p->g.nrows * p->g.P * sizeof(double)));
        p->allocs.push_back(q);
        p->tmp[i] = (double *)q;
    }
    *base = p->tmp[i];
    return 0;
}

static int g_opt_f64_pairs = getenv("WT_NO_F64_PAIRS") ? 0 : 1;      // wt_set_option("f64_pairs", 0/1): two pixels per thread
void wt_set_f64_pairs(int on) { g_opt_f64_pairs = on; }
static inline wt_ctx *ctx_of(wt_plan64 *p) { return p ? p->ctx : nullptr; }
static inline dim3 grid64(const wt_plan64 *p) { return dim3((p->g.W + 255) / 256, (unsigned)std::min(p->g.nrows, 32768)); }
static inline dim3 grid64_pairs(const wt_plan64 *p) { return dim3((p->g.W / 2 + 255) / 256, (unsigned)std::min(p->g.nrows, 32768)); }
// flat pointwise kernels: planes are contiguous (pitch even, 16-byte aligned rows): double2 groups of a plane
static inline int64_t plan64_n2(const wt_plan64 *p) { return (int64_t)p->g.nrows * p->g.P / 2; }
static inline int flat_grid64(const wt_plan64 *p) { return (int)std::min<int64_t>((plan64_n2(p) + 255) / 256, 256 * 16); }
static Taps64 taps64(const wt_plan64 *p)
{
    Taps64 t{};
    t.n = p->ntaps;
    for (int i = 0; i < p->ntaps; ++i) t.k[i] = p->taps[i];
    return t;
}

extern "C" int wt64_plan_create(wt_ctx *ctx, int64_t H, int64_t W, int max_level, const double *taps, int ntaps, wt_plan64 **out)
{
    WtGuard guard_(ctx);
    if (!ctx || !out || !taps) WT_FAIL("wt64_plan_create: null pointer");
    if (H < 1 || W < 1 || H > (1 << 30) || W > (1 << 30)) WT_FAIL("wt64_plan_create: bad image size %lld x %lld", (long long)H, (long long)W);
    if (max_level < 0 || max_level > 30) WT_FAIL("wt64_plan_create: max_level %d out of range", max_level);
    if (ntaps < 1 || ntaps > WT64_MAX_TAPS || !(ntaps & 1)) WT_FAIL("wt64_plan_create: %d taps unsupported (odd, 1..%d)", ntaps, WT64_MAX_TAPS);
    if ((int64_t)H * ((W + 1) / 2 * 2) > ((int64_t)1 << 31) - 1) WT_FAIL("wt64_plan_create: planes beyond 2^31 samples are not supported in float64");
    wt_plan64 *p = new wt_plan64();
    p->ctx = ctx;
    p->g.W = (int)W;
    p->g.P = (int)((W + 1) / 2 * 2);
    p->g.H = (int)H;
    p->g.row0 = 0;
    p->g.nrows = (int)H;
    p->g.halo = 0;
    p->g.border = 0;
    p->max_level = max_level;
    p->coef.assign(max_level + 1, nullptr);
    p->ntaps = ntaps;
    for (int i = 0; i < ntaps; ++i) p->taps[i] = taps[i];
    *out = p;
    return 0;
}

extern "C" int wt64_plan_destroy(wt_plan64 *p)
{
    WtGuard guard_(ctx_of(p));
    if (!p) return 0;
    if (p->ctx->prehist_plan == p) p->ctx->prehist_plan = nullptr;
    (void)hipSetDevice(p->ctx->device);
    (void)wt_side_join(p->ctx);
    (void)hipStreamSynchronize(p->ctx->stream);
    destroy_events(p->scale_ev);
    for (void *q : p->allocs) (void)hipFree(q);
    delete p;
    return 0;
}

/* border modes 0..3 as wt_plan_set_border (2 / 3: the 'mirror' border of 1-D signals, 1 x N images) */
extern "C" int wt64_plan_set_border(wt_plan64 *p, int border)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_plan_set_border: null plan");
    if (border < 0 || border > 3) WT_FAIL("wt64_plan_set_border: border %d unsupported (0..3, as wt_plan_set_border)", border);
    p->g.border = border;
    return 0;
}

extern "C" int wt64_upload(wt_plan64 *p, int plane, const double *host, int64_t host_pitch)
{
    WtGuard guard_(ctx_of(p));
    if (!p || !host) WT_FAIL("wt64_upload: null pointer");
    double *b = nullptr;
    WT_TRY(plan64_base(p, plane, &b));
    WT_HIP(hipMemcpy2DAsync(b, (size_t)p->g.P * 8, host, (size_t)host_pitch * 8, (size_t)p->g.W * 8, p->g.nrows, hipMemcpyHostToDevice, p->ctx->stream));
    WT_HIP(hipStreamSynchronize(p->ctx->stream));
    return 0;
}

// Integer images (watroo/wavelets.py:297, 319-320: the reference promotes them to float64 on the host - a
// numpy astype of 30 ms for a 4096^2 int16 frame, and four times the bytes over PCIe).  Here the integers
// cross as they are and one kernel widens them into the plane; int -> double is exact up to 2^53 and
// rounds to nearest even beyond, as numpy's astype does.
template <typename I>
static void wt64_from_int_launch(wt_plan64 *p, double *b, bool swap)
{
    const dim3 grid = grid64(p), block(256);
    if (swap) hipLaunchKernelGGL((wt_from_elems_kernel<I, double, true>), grid, block, 0, p->ctx->stream, (const I *)p->istage, b, p->g.W, p->g.P, p->g.nrows);
    else hipLaunchKernelGGL((wt_from_elems_kernel<I, double, false>), grid, block, 0, p->ctx->stream, (const I *)p->istage, b, p->g.W, p->g.P, p->g.nrows);
}

extern "C" int wt64_upload_int(wt_plan64 *p, int plane, const void *host, int64_t host_pitch_bytes, int dtype)
{
    WtGuard guard_(ctx_of(p));
    if (!p || !host) WT_FAIL("wt64_upload_int: null pointer");
    static const int isz[11] = {0, 1, 1, 2, 2, 4, 4, 8, 8, 4, 8};
    const bool swap = (dtype & WT_BYTESWAPPED) != 0;
    const int base = dtype & ~WT_BYTESWAPPED;
    if (base < WT_INT8 || base > WT_FLOAT64) WT_FAIL("wt64_upload_int: unknown element type %d", dtype);
    const size_t row = (size_t)p->g.W * isz[base];
    if (host_pitch_bytes < (int64_t)row) WT_FAIL("wt64_upload_int: row pitch %lld below the %zu bytes of a row", (long long)host_pitch_bytes, row);
    double *b = nullptr;
    WT_TRY(plan64_base(p, plane, &b));
    const size_t need = row * p->g.nrows;
    if (p->istage_cap < need) {
        WT_HIP(hipSetDevice(p->ctx->device));
        WT_HIP(hipStreamSynchronize(p->ctx->stream));
        if (p->istage) {
            (void)hipFree(p->istage);
            p->allocs.erase(std::remove(p->allocs.begin(), p->allocs.end(), p->istage), p->allocs.end());
            p->istage = nullptr;
            p->istage_cap = 0;
        }
        WT_HIP(hipMalloc(&p->istage, need));
        p->allocs.push_back(p->istage);
        p->istage_cap = need;
    }
    const bool pinned = try_pin(host, (size_t)(p->g.nrows - 1) * (size_t)host_pitch_bytes + row);
    hipError_t e = hipMemcpy2DAsync(p->istage, row, host, (size_t)host_pitch_bytes, row, p->g.nrows, hipMemcpyHostToDevice, p->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->ctx->stream);      // (the host rows are free again)
    if (pinned) (void)hipHostUnregister(const_cast<void *>(host));
    WT_HIP(e);
    switch (base) {
        case WT_INT8: wt64_from_int_launch<int8_t>(p, b, false); break;
        case WT_UINT8: wt64_from_int_launch<uint8_t>(p, b, false); break;
        case WT_INT16: wt64_from_int_launch<int16_t>(p, b, swap); break;
        case WT_UINT16: wt64_from_int_launch<uint16_t>(p, b, swap); break;
        case WT_INT32: wt64_from_int_launch<int32_t>(p, b, swap); break;
        case WT_UINT32: wt64_from_int_launch<uint32_t>(p, b, swap); break;
        case WT_INT64: wt64_from_int_launch<int64_t>(p, b, swap); break;
        case WT_UINT64: wt64_from_int_launch<uint64_t>(p, b, swap); break;
        case WT_FLOAT32: wt64_from_int_launch<float>(p, b, swap); break;
        default: wt64_from_int_launch<double>(p, b, swap); break;
    }
    WT_HIP(hipGetLastError());
    WT_HIP(hipStreamSynchronize(p->ctx->stream));          // the caller's buffer is free again when this returns (as wt64_upload)
    return 0;
}

extern "C" int wt64_download(wt_plan64 *p, int plane, double *host, int64_t host_pitch)
{
    WtGuard guard_(ctx_of(p));
    if (!p || !host) WT_FAIL("wt64_download: null pointer");
    double *b = nullptr;
    WT_TRY(plan64_base(p, plane, &b));
    WT_HIP(hipMemcpy2DAsync(host, (size_t)host_pitch * 8, b, (size_t)p->g.P * 8, (size_t)p->g.W * 8, p->g.nrows, hipMemcpyDeviceToHost, p->ctx->stream));
    WT_HIP(hipStreamSynchronize(p->ctx->stream));
    return 0;
}

// Images whose taps are one of the built-in families run the per-scale kernels of wt_stencil.h instantiated
// for double (round 5: lattice / row / chain kernels, the fused wow update, the bilateral march) instead of
// the generic one-sample-per-thread kernels of this file; wt_set_option("stencil64", 0) restores those (A/B,
// tests: both orders of operations are rows first, then columns, FMA chains in tap order - identical bits
// for the filters).
static int fused64_family(const wt_plan64 *p);
static int g_opt_stencil64 = getenv("WT_NO_STENCIL64") ? 0 : 1;
void wt_set_stencil64(int on) { g_opt_stencil64 = on; }
static inline bool stencil64_ok(const wt_plan64 *p) { return g_opt_stencil64 && fused64_family(p) >= 0 && p->g.H >= 2; }
static inline StencilCtx stencil64_ctx(const wt_plan64 *p, hipStream_t st = nullptr)
{
    return StencilCtx{p->ctx, st ? st : p->ctx->stream, p->g, fused64_family(p)};
}

// conv_s of a plane: rows -> a private temporary, axis 1 (-> a second one when an axis-0 pass follows), axis 0.
// depth = 0: an image (or a 1 x N signal: no column pass); depth = Z > 0: a (Z, Y, X) cube.
static int smooth64(wt_plan64 *p, const double *in, double *out, double *out_w, int s, int square, int depth)
{
    if (s < 0 || s > 24) WT_FAIL("float64 plan: scale %d out of range", s);
    if (depth < 0 || (depth > 0 && p->g.H % depth)) WT_FAIL("float64 plan: height %d is not a multiple of depth %d", p->g.H, depth);
    if (out == in || out_w == in) WT_FAIL("float64 plan: in-place filtering");
    const Taps64 t = taps64(p);
    const int d = 1 << s;
    const Geo g = p->g;
    const dim3 grid = grid64(p), block(256);
    const int Z = depth > 0 ? depth : 1, Y = g.H / Z;
    // a 1 x N image under the 'mirror' border is a 1-D signal (watroo/wavelets.py:65-69: row filter
    // only); any other one-row image or one-slice cube still sees every axis' taps (they reflect
    // onto the same sample and contribute sum(k) - which is 1 only for normalised taps)
    const bool cols = !(Y == 1 && depth == 0 && (g.border == 2 || g.border == 3)), deep = depth > 0;
    // One marching kernel or the two plain passes (identical bits)?  Measured per scale (ms, B3, chain | two
    // passes): 2048^2 0.076 | 0.059 at every d; 4096^2 0.24 | 0.24 up to d = 64, 0.31 | 0.29 at 256, 0.64 | 0.28
    // at 1024; 8192^2 0.90 | 1.03 up to d = 128, 1.03 | 1.18 at 256, 1.23 | 1.12 at 512, 1.66 | 1.19 at 1024.
    // The chain saves a plane round trip - which only costs when the planes do not sit in the Infinity
    // Cache - and loses when the chains get short (2 * hw warm-up rows per chunk, K scalar loads per row).
    static const int chain_env = getenv("WT64_CHAIN") ? atoi(getenv("WT64_CHAIN")) : -1;       // 0 / 1 force (experiments)
    const bool big_planes = (size_t)g.nrows * g.P * sizeof(double) >= ((size_t)256 << 20);
    const bool use_chain = chain_env >= 0 ? chain_env != 0 : (big_planes && (g.H + d - 1) / d >= 24);
    if (cols && !deep && stencil64_ok(p)) {
        // images, built-in taps: one tiled kernel per scale (wt_stencil.h)
        ChainArgsT<double> a{};
        a.in = in; a.out_c = out; a.out_w = out_w;
        a.f1 = 1.0; a.f2 = 1.0;
        return wt64_stencil_launch(stencil64_ctx(p), square ? MODE_SMOOTH_SQ : (out_w ? MODE_DECOMP : MODE_SMOOTH), a, s);
    }
    if (cols && !deep && !square && use_chain) {
        // images: one kernel per scale (register window down every polyphase row chain)
        const int n_max = (g.H + d - 1) / d;             // longest chain
        // enough work items to fill the chip, chunks of at least 32 chain steps (2 * hw warm-up rows each)
        const int xblocks = (g.W + 255) / 256;
        int chunks = std::max(1, std::min((n_max + 31) / 32, (4 * p->ctx->num_cus + xblocks * d - 1) / (xblocks * d)));
        const int S = (n_max + chunks - 1) / chunks;
        chunks = (n_max + S - 1) / S;
        const dim3 cgrid(xblocks, (unsigned)std::min<int64_t>((int64_t)d * chunks, 65535));
        hipLaunchKernelGGL(wt64_chain_kernel, cgrid, block, 0, p->ctx->stream, in, out, out_w, g, d, t, S, chunks);
        WT_HIP(hipGetLastError());
        return 0;
    }
    double *t1 = nullptr, *t2 = nullptr;                 // private temporaries, allocated on first use
    if (cols || deep) WT_TRY(plan64_tmp(p, 0, &t1));
    if (cols && deep) WT_TRY(plan64_tmp(p, 1, &t2));
    double *r_out = (cols || deep) ? t1 : out;
    const bool pairs = g_opt_f64_pairs && g.W % 2 == 0;              // two pixels per thread (16-byte accesses)
    if (pairs && d % 2 == 0) hipLaunchKernelGGL(wt64_rows2_kernel, grid64_pairs(p), block, 0, p->ctx->stream, in, r_out, g, d, t, square);
    else hipLaunchKernelGGL(wt64_rows_kernel, grid, block, 0, p->ctx->stream, in, r_out, g, d, t, square);
    if (!cols && !deep) {
        if (out_w) hipLaunchKernelGGL(wt64_binary_kernel, grid, block, 0, p->ctx->stream, in, (const double *)out, out_w, g.W, g.P, g.nrows, 1);
    } else if (cols && !deep && pairs) {
        hipLaunchKernelGGL(wt64_cols2_kernel, grid64_pairs(p), block, 0, p->ctx->stream, (const double *)t1, out, in, out_w, g.W, g.P, g.nrows, d,
                           g.border, t);
    } else if (cols) {
        double *c_out = deep ? t2 : out;
        hipLaunchKernelGGL(wt64_axis_kernel, grid, block, 0, p->ctx->stream, (const double *)t1, c_out, in, deep ? (double *)nullptr : out_w,
                           g.W, g.P, Y, Z, d, g.border, t, 1);
        if (deep)
            hipLaunchKernelGGL(wt64_axis_kernel, grid, block, 0, p->ctx->stream, (const double *)t2, out, in, out_w, g.W, g.P, Y, Z, d, g.border, t, 0);
    } else {
        hipLaunchKernelGGL(wt64_axis_kernel, grid, block, 0, p->ctx->stream, (const double *)t1, out, in, out_w, g.W, g.P, Y, Z, d, g.border, t, 0);
    }
    WT_HIP(hipGetLastError());
    return 0;
}

/* convolution(arr, scaling_function, s) in float64 (watroo/wavelets.py:35-69); square_input as wt_smooth */
extern "C" int wt64_smooth(wt_plan64 *p, int src, int dst, int s, int square_input, int depth)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_smooth: null plan");
    double *in = nullptr, *o = nullptr;
    WT_TRY(plan64_base(p, src, &in));
    WT_TRY(plan64_base(p, dst, &o));
    return smooth64(p, in, o, nullptr, s, square_input, depth);
}

// ---------------------------------------------------------------------------------------------
// fused multi-scale passes in float64 (round 3): wt_fused_kernel instantiated for double - a lane
// owns two pixels (double2 = the same 16 bytes per access and the same register window as the
// float4 passes), same polyphase march, LDS row exchange, delay rings and fixed-descriptor stores.
// Served: images (H >= 2) under the symmetric border whose taps are one of the built-in families and
// whose schedule is all fused passes; signals, cubes, user-defined taps, the borders of the recursive
// algorithm and schedules with per-scale passes (L > 8) keep the generic per-scale kernels above.
// ---------------------------------------------------------------------------------------------
static int g_opt_fused64 = getenv("WT_NO_FUSED64") ? 0 : 1;      // wt_set_option("fused64", 0/1)
void wt_set_fused64(int on) { g_opt_fused64 = on; }

static int fused64_family(const wt_plan64 *p)
{
    static const double b3[5] = {1. / 16, 1. / 4, 3. / 8, 1. / 4, 1. / 16}, tri[3] = {1. / 4, 1. / 2, 1. / 4};
    if (p->ntaps == 5 && !memcmp(p->taps, b3, sizeof b3)) return WT_B3SPLINE;
    if (p->ntaps == 3 && !memcmp(p->taps, tri, sizeof tri)) return WT_TRIANGLE;
    return -1;
}

// every pass of the fused schedule for `level` scales exists (any width and height: images that do
// not admit the fast addressing - odd widths, sizes below a pass's halo - take the generic one, as in
// float32)
static bool fused64_ok(const wt_plan64 *p, int level, int depth, int32_t *tr, int *np, bool *all_fused)
{
    const int fam = fused64_family(p);
    *all_fused = false;
    if (!g_opt_fused64 || fam < 0 || depth != 0 || p->g.border != 0 || level < 1) return false;
    if (p->g.H < 2) return false;                        // 1 x N: a signal (row filter only) - generic kernels
    if (!wt_fused_supported_bytes((int64_t)p->g.P * 8)) return false;
    if (wt_schedule(fam, level, 1, tr, 32, np)) return false;
    bool all = true, any = false;
    for (int i = 0; i < *np; ++i) {
        const bool has = wt_fused_has_pass(tr[3 * i], tr[3 * i + 1], fam);
        all = all && has;
        any = any || has;
    }
    *all_fused = all;
    return any;                                          // (scales beyond the fused passes: one generic kernel each)
}

// one fused pass: scales [s0, s0 + ns) of plane cur -> detail planes s0.. and plane nxt; acc 0 plain,
// 1 / 2 carrying the plane sum in sum_plane (2: last pass), 3 plain + first level of the median select
static int fused64_pass(wt_plan64 *p, int cur, int nxt, int s0, int ns, int acc, bool first_of_sum, int sum_plane)
{
    const int fam = fused64_family(p);
    if (fam < 0 || !wt_fused_has_pass(s0, ns, fam)) WT_FAIL("float64 pass (%d, %d): no fused kernel for these taps / scales", s0, ns);
    if (p->g.border != 0 || p->g.H < 2 || !wt_fused_supported_bytes((int64_t)p->g.P * 8)) WT_FAIL("float64 pass: the fused passes need an image under the symmetric border");
    if (s0 < 0 || s0 + ns - 1 > p->max_level || ns < 1) WT_FAIL("float64 pass: scales [%d,%d) outside the plan (max_level %d)", s0, s0 + ns, p->max_level);
    if (cur == nxt || (cur >= s0 && cur < s0 + ns) || (nxt >= s0 && nxt < s0 + ns)) WT_FAIL("float64 pass: input / output planes alias the detail planes of the pass");
    const bool b3 = fam == WT_B3SPLINE;
    FusedArgsT<double> a{};
    double *in = nullptr;
    WT_TRY(plan64_base(p, cur, &in));
    a.in = in;
    WT_TRY(plan64_base(p, nxt, &a.out_c));
    for (int k = 0; k < ns && k < 3; ++k) WT_TRY(plan64_base(p, s0 + k, &a.out_w[k]));
    if (ns > 3) WT_TRY(plan64_base(p, s0 + 3, &a.out_w3));
    a.g = p->g;
    if (acc == 1 || acc == 2) {
        if (first_of_sum != (s0 == 0)) WT_FAIL("float64 pass: first must be set for the pass that starts at scale 0 and only for it");
        WT_TRY(plan64_base(p, sum_plane, &a.p_out));
        a.p_in = first_of_sum ? nullptr : a.p_out;
    }
    const FusedRows rows;          // (whole passes: the float64 engine is single-GPU)
    if (acc == 3) {
        if (s0 != 0) WT_FAIL("float64 pass: the histogram variant exists for the first pass only");
        a.hist = p->ctx->d_hist;
        a.hist_base = p->ctx->prehist_windowed ? hist_base_word(p->ctx) : nullptr;
        p->ctx->prehist_ran = true;
        return b3 ? wt_fused_tu_f64_k5_acc3(p, a, s0, ns, rows) : wt_fused_tu_f64_k3_acc3(p, a, s0, ns, rows);
    }
    if (acc == 2) return b3 ? wt_fused_tu_f64_k5_acc2(p, a, s0, ns, rows) : wt_fused_tu_f64_k3_acc2(p, a, s0, ns, rows);
    if (acc == 1) return b3 ? wt_fused_tu_f64_k5_acc1(p, a, s0, ns, rows) : wt_fused_tu_f64_k3_acc1(p, a, s0, ns, rows);
    return b3 ? wt_fused_tu_f64_k5_acc0(p, a, s0, ns, rows) : wt_fused_tu_f64_k3_acc0(p, a, s0, ns, rows);
}

// flag bit4 of wt64_decompose_ex / wt64_decompose_pass: the first pass also histograms the exponent
// field of |w_0| (first level of wt64_abs_median's select); same marker protocol as the float32 engine
static int prehist64_begin(wt_plan64 *p, int flags, int src = WT_PLANE_NONE)
{
    wt_ctx *c = p->ctx;
    c->prehist_ran = false;
    if (flags & 16) {
        c->prehist_plan = nullptr;
        c->prehist_windowed = false;
        WT_HIP(hipMemsetAsync(c->d_hist, 0, WT_HIST_BINS * sizeof(uint32_t), c->stream));
        // the windowed form, as in float32 (prehist_begin): 22-bit keys around a median predicted from
        // 4096 pixels of `src` - the select then goes straight to its gather pass
        const int fam = fused64_family(p);
        if (g_opt_hist_window && src != WT_PLANE_NONE && fam >= 0 && !p->g.border && (int64_t)p->g.H * p->g.W >= ((int64_t)1 << 20) &&
            p->g.H >= 64 && p->g.W >= 64) {
            double *in = nullptr;
            WT_TRY(plan64_base(p, src, &in));
            ProfScope ps(c, "wt_median_window_kernel");
            uint32_t *keys = (uint32_t *)c->d_partials;
            if (fam == WT_B3SPLINE) hipLaunchKernelGGL((wt_median_sample_kernel<5, double>), dim3(64), dim3(64), 0, c->stream, (const double *)in, p->g, keys);
            else hipLaunchKernelGGL((wt_median_sample_kernel<3, double>), dim3(64), dim3(64), 0, c->stream, (const double *)in, p->g, keys);
            hipLaunchKernelGGL(wt_median_window_kernel, dim3(1), dim3(1024), 0, c->stream, (const uint32_t *)keys, hist_base_word(c));
            WT_HIP(hipGetLastError());
            c->prehist_windowed = true;
        }
    }
    return 0;
}
static void prehist64_end(wt_plan64 *p)
{
    if (p->ctx->prehist_ran) {
        p->ctx->prehist_plan = p;
        p->ctx->prehist_plane = 0;
    }
    p->ctx->prehist_ran = false;
}

// the fused schedule: planes 0..level, optionally the plane sum carried through the passes into dst
static int fused64_run(wt_plan64 *p, int src, int level, const int32_t *tr, int np, bool with_sum, int dst, int flags = 0)
{
    const bool b3 = fused64_family(p) == WT_B3SPLINE;
    int cur = src;
    for (int i = 0; i < np; ++i) {
        const int s0 = tr[3 * i], ns = tr[3 * i + 1];
        const bool last = s0 + ns == level;
        const int nxt = last ? level : WT_PLANE_SCRATCH(i & 1);
        if (!wt_fused_has_pass(s0, ns, b3 ? WT_B3SPLINE : WT_TRIANGLE)) {
            // a scale beyond the fused passes (the schedule gives those one scale per pass)
            if (with_sum || ns != 1) WT_FAIL("float64 schedule: pass (%d, %d) has no fused kernel", s0, ns);
            double *ci = nullptr, *co = nullptr, *w = nullptr;
            WT_TRY(plan64_base(p, cur, &ci));
            WT_TRY(plan64_base(p, nxt, &co));
            WT_TRY(plan64_base(p, s0, &w));
            WT_TRY(smooth64(p, ci, co, w, s0, 0, 0));
            cur = nxt;
            continue;
        }
        const int acc = with_sum ? (last ? 2 : 1) : ((flags & 16) && s0 == 0 ? 3 : 0);
        WT_TRY(fused64_pass(p, cur, nxt, s0, ns, acc, i == 0, dst));
        cur = nxt;
    }
    return 0;
}

/* AtrousTransform.atrous_standard in float64 (watroo/wavelets.py:408-444): planes 0..level-1 detail,
 * plane level smooth, from plane src (left intact).  depth as above. */
extern "C" int wt64_decompose_ex(wt_plan64 *p, int src, int level, int depth, int flags);
extern "C" int wt64_decompose(wt_plan64 *p, int src, int level, int depth)
{
    WtGuard guard_(ctx_of(p));
    return wt64_decompose_ex(p, src, level, depth, 0);
}

/* wt64_decompose with flags: bit4 (16) = where the first pass is a fused one it also histograms the
 * first radix level of |w_0| for a wt64_abs_median(plan, 0) that follows (as flag bit4 of wt_decompose) */
extern "C" int wt64_decompose_ex(wt_plan64 *p, int src, int level, int depth, int flags)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_decompose: null plan");
    if (level < 0 || level > p->max_level) WT_FAIL("wt64_decompose: level %d exceeds plan max_level %d", level, p->max_level);
    if (src >= 0 && src <= level) WT_FAIL("wt64_decompose: src plane %d is one of the output planes", src);
    if (src == WT_PLANE_SCRATCH(0) || src == WT_PLANE_SCRATCH(1)) WT_FAIL("wt64_decompose: scratch planes 0/1 are used internally");
    double *in = nullptr;
    WT_TRY(plan64_base(p, src, &in));
    if (level == 0) {
        double *o = nullptr;
        WT_TRY(plan64_base(p, 0, &o));
        WT_HIP(hipMemcpyAsync(o, in, (size_t)p->g.nrows * p->g.P * 8, hipMemcpyDeviceToDevice, p->ctx->stream));
        return 0;
    }
    {
        int32_t tr[3 * 32];
        int np = 0;
        bool all = false;
        if (fused64_ok(p, level, depth, tr, &np, &all)) {
            WT_TRY(prehist64_begin(p, flags, src));
            WT_TRY(fused64_run(p, src, level, tr, np, false, WT_PLANE_NONE, flags));
            prehist64_end(p);
            return 0;
        }
    }
    int cur = src;
    for (int s = 0; s < level; ++s) {
        const int nxt = (s == level - 1) ? level : WT_PLANE_SCRATCH(s & 1);
        double *ci = nullptr, *co = nullptr, *w = nullptr;
        WT_TRY(plan64_base(p, cur, &ci));
        WT_TRY(plan64_base(p, nxt, &co));
        WT_TRY(plan64_base(p, s, &w));
        WT_TRY(smooth64(p, ci, co, w, s, 0, depth));
        cur = nxt;
    }
    return 0;
}

extern "C" int wt64_plane_sum(wt_plan64 *p, int first, int count, int dst);

/* wt_decompose_sum in float64: the transform and np.sum(planes, axis=0) -> dst; the sum rides in the
 * fused passes where the schedule is fused (plane order, the association of numpy's sum over axis
 * 0), else the two-step form.  *fused_out (may be null) tells which. */
extern "C" int wt64_decompose_sum(wt_plan64 *p, int src, int level, int dst, int *fused_out)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_decompose_sum: null plan");
    if (level < 0 || level > p->max_level) WT_FAIL("wt64_decompose_sum: level %d exceeds plan max_level %d", level, p->max_level);
    if ((src >= 0 && src <= level) || (dst >= 0 && dst <= level) || dst == src)
        WT_FAIL("wt64_decompose_sum: src / dst plane is one of the output planes (or the same plane)");
    if (src == WT_PLANE_SCRATCH(0) || src == WT_PLANE_SCRATCH(1) || dst == WT_PLANE_SCRATCH(0) || dst == WT_PLANE_SCRATCH(1))
        WT_FAIL("wt64_decompose_sum: scratch planes 0/1 are used internally");
    int32_t tr[3 * 32];
    int np = 0;
    bool all = false;
    const bool fused = fused64_ok(p, level, 0, tr, &np, &all) && all;    // the sum rides only if every pass is fused
    if (fused_out) *fused_out = fused ? 1 : 0;
    if (fused) return fused64_run(p, src, level, tr, np, true, dst);
    WT_TRY(wt64_decompose(p, src, level, 0));
    return wt64_plane_sum(p, 0, level + 1, dst);
}

/* Would wt64_decompose_sum(plan, ., level, .) carry the sum through fused passes (wt_plan_fused_ok)? */
extern "C" int wt64_plan_fused_ok(wt_plan64 *p, int level, int *ok)
{
    WtGuard guard_(ctx_of(p));
    if (!p || !ok) WT_FAIL("wt64_plan_fused_ok: null pointer");
    int32_t tr[3 * 32];
    int np = 0;
    bool all = false;
    *ok = (level >= 1 && level <= p->max_level && fused64_ok(p, level, 0, tr, &np, &all) && all) ? 1 : 0;
    return 0;
}

/* One pass of the fused schedule in float64 (wt_decompose_pass): scales [s0, s0 + ns) of plane cur ->
 * detail planes s0 .. s0 + ns - 1 and plane nxt; flags bit4: also histogram |w_0| (s0 = 0 only) */
extern "C" int wt64_decompose_pass(wt_plan64 *p, int cur, int nxt, int s0, int ns, int flags)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_decompose_pass: null plan");
    WT_TRY(prehist64_begin(p, flags, s0 == 0 ? cur : WT_PLANE_NONE));
    WT_TRY(fused64_pass(p, cur, nxt, s0, ns, (flags & 16) && s0 == 0 ? 3 : 0, false, WT_PLANE_NONE));
    prehist64_end(p);
    return 0;
}

/* wt_decompose_pass_sum in float64: the pass also carries np.sum(planes, axis=0) in sum_plane */
extern "C" int wt64_decompose_pass_sum(wt_plan64 *p, int cur, int nxt, int s0, int ns, int sum_plane, int first, int last)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_decompose_pass_sum: null plan");
    if (sum_plane == cur || sum_plane == nxt || (sum_plane >= s0 && sum_plane < s0 + ns)) WT_FAIL("wt64_decompose_pass_sum: the sum plane aliases a plane of the pass");
    return fused64_pass(p, cur, nxt, s0, ns, last ? 2 : 1, first != 0, sum_plane);
}

/* sdev_loc(image, sf, s, variance) (watroo/wavelets.py:24-32), times f1 then f2 */
extern "C" int wt64_local_variance(wt_plan64 *p, int src, int dst, int s, double f1, double f2, int take_sqrt, int depth)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_local_variance: null plan");
    if (src == dst) WT_FAIL("wt64_local_variance: src and dst must differ");
    double *in = nullptr, *o = nullptr, *mean = nullptr;
    WT_TRY(plan64_base(p, src, &in));
    WT_TRY(plan64_base(p, dst, &o));
    if (depth == 0 && stencil64_ok(p) && !(p->g.H == 1 && (p->g.border == 2 || p->g.border == 3))) {
        // images, built-in taps: both moments in one marching kernel (wt_stencil.h, MODE_VAR)
        ChainArgsT<double> a{};
        a.in = in; a.out_c = o;
        a.f1 = f1; a.f2 = f2; a.take_sqrt = take_sqrt;
        return wt64_stencil_launch(stencil64_ctx(p), MODE_VAR, a, s);
    }
    WT_TRY(plan64_tmp(p, 2, &mean));
    WT_TRY(smooth64(p, in, mean, nullptr, s, 0, depth));
    WT_TRY(smooth64(p, in, o, nullptr, s, 1, depth));
    hipLaunchKernelGGL(wt64_var_kernel, grid64(p), dim3(256), 0, p->ctx->stream, (const double *)mean, (const double *)o, o, p->g.W, p->g.P, p->g.nrows,
                       f1, f2, take_sqrt);
    WT_HIP(hipGetLastError());
    return 0;
}

extern "C" int wt64_binary(wt_plan64 *p, int op, int a, int b, int dst)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_binary: null plan");
    if (op < 0 || op > 4) WT_FAIL("wt64_binary: unknown op %d", op);
    double *pa = nullptr, *pb = nullptr, *pd = nullptr;
    WT_TRY(plan64_base(p, a, &pa));
    WT_TRY(plan64_base(p, b, &pb));
    WT_TRY(plan64_base(p, dst, &pd));
    hipLaunchKernelGGL(wt64_binary_kernel, grid64(p), dim3(256), 0, p->ctx->stream, (const double *)pa, (const double *)pb, pd, p->g.W, p->g.P, p->g.nrows, op);
    WT_HIP(hipGetLastError());
    return 0;
}

/* mode 0: Coefficients.significance into dst; mode 1: dst = src * (wgt * significance)
 * (Coefficients.denoise with dst == src), watroo/wavelets.py:129-149 */
extern "C" int wt64_significance(wt_plan64 *p, int src, int dst, double tau, double wgt, int soft, int noise_plane, int mode)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_significance: null plan");
    double *c = nullptr, *d = nullptr, *nz = nullptr;
    WT_TRY(plan64_base(p, src, &c));
    WT_TRY(plan64_base(p, dst, &d));
    if (noise_plane != WT_PLANE_NONE) WT_TRY(plan64_base(p, noise_plane, &nz));
    ProfScope ps(p->ctx, "wt64_signif_kernel");
    hipLaunchKernelGGL(wt64_signif_kernel, dim3(flat_grid64(p)), dim3(256), 0, p->ctx->stream, (const double *)c, (const double *)nz, d, plan64_n2(p), tau,
                       wgt, soft, mode);
    WT_HIP(hipGetLastError());
    return 0;
}

extern "C" int wt64_plane_sum(wt_plan64 *p, int first, int count, int dst)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_plane_sum: null plan");
    if (count < 1 || count > 32 || first < 0 || first + count - 1 > p->max_level) WT_FAIL("wt64_plane_sum: planes [%d,%d) outside [0,%d]", first, first + count, p->max_level);
    Sum64Args a{};
    a.n = count;
    for (int i = 0; i < count; ++i) {
        double *b = nullptr;
        WT_TRY(plan64_base(p, first + i, &b));
        a.p[i] = b;
    }
    double *d = nullptr;
    WT_TRY(plan64_base(p, dst, &d));
    ProfScope ps(p->ctx, "wt64_plane_sum_kernel");
    hipLaunchKernelGGL(wt64_plane_sum_kernel, dim3((unsigned)((plan64_n2(p) + 255) / 256)), dim3(256), 0, p->ctx->stream, a, d, plan64_n2(p));
    WT_HIP(hipGetLastError());
    return 0;
}

// the two-part sum of wow() behind a bilateral transform: see wt_plane_sum_early (wt_apps.hip)
static int plane_sum64_launch(wt_plan64 *p, const double *acc, int first, int count, double *d)
{
    Sum64Args a{};
    a.n = 0;
    if (acc) a.p[a.n++] = acc;
    for (int i = 0; i < count; ++i) {
        double *b = nullptr;
        WT_TRY(plan64_base(p, first + i, &b));
        a.p[a.n++] = b;
    }
    ProfScope ps(p->ctx, "wt64_plane_sum_kernel");
    hipLaunchKernelGGL(wt64_plane_sum_kernel, dim3((unsigned)((plan64_n2(p) + 255) / 256)), dim3(256), 0, p->ctx->stream, a, d, plan64_n2(p));
    WT_HIP(hipGetLastError());
    return 0;
}

extern "C" int wt64_plane_sum_early(wt_plan64 *p, int count, int dst, int *done)
{
    WtGuard guard_(ctx_of(p));
    if (!p || !done) WT_FAIL("wt64_plane_sum_early: null pointer");
    *done = 0;
    if (count < 1 || count > 32 || count - 1 > p->max_level) WT_FAIL("wt64_plane_sum_early: %d planes outside [1,%d]", count, p->max_level + 1);
    if (dst >= 0) WT_FAIL("wt64_plane_sum_early: dst must not be a coefficient plane");
    const bool side = wt_wow_overlap_enabled() && p->overlap_ok && count <= p->overlap_scales && p->ctx->side_pending;
    if (!side) return 0;
    WtSideScope side_scope(p->ctx, p->scale_ev[count - 1], true);
    if (!side_scope.ok()) return 2;
    double *d = nullptr;
    WT_TRY(plan64_base(p, dst, &d));
    WT_TRY(plane_sum64_launch(p, nullptr, 0, count, d));
    *done = 1;
    return 0;
}

extern "C" int wt64_plane_sum_resume(wt_plan64 *p, int first, int count, int dst)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_plane_sum_resume: null plan");
    if (count < 1 || count + 1 > 32 || first < 0 || first + count - 1 > p->max_level)
        WT_FAIL("wt64_plane_sum_resume: planes [%d,%d) outside [0,%d]", first, first + count, p->max_level);
    if (dst >= 0) WT_FAIL("wt64_plane_sum_resume: dst must not be a coefficient plane");
    double *d = nullptr;
    WT_TRY(plan64_base(p, dst, &d));
    return plane_sum64_launch(p, d, first, count, d);
}

/* wt_denoise_sum in float64 (Coefficients.denoise over the first n_den planes + np.sum(planes, axis=0),
 * watroo/wavelets.py:145-149, utils.py:98): planes [first, first + count) -> dst; tau[k] <= 0: no
 * threshold on plane k (its weight still applies); write_back: store the thresholded planes */
extern "C" int wt64_denoise_sum(wt_plan64 *p, int first, int count, int dst, int n_den, const double *tau, const double *wgt, int soft,
                                int noise_plane, int write_back)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_denoise_sum: null plan");
    if (count < 1 || count > 32 || first < 0 || first + count - 1 > p->max_level) WT_FAIL("wt64_denoise_sum: planes [%d,%d) outside [0,%d]", first, first + count, p->max_level);
    if (n_den < 0 || n_den > count) WT_FAIL("wt64_denoise_sum: n_den %d outside [0,%d]", n_den, count);
    if (n_den > 0 && (!tau || !wgt)) WT_FAIL("wt64_denoise_sum: null tau / wgt");
    DenoiseSum64Args a{};
    a.n = count; a.n_den = n_den; a.soft = soft; a.write_back = write_back;
    for (int i = 0; i < count; ++i) {
        WT_TRY(plan64_base(p, first + i, &a.p[i]));
        a.tau[i] = i < n_den ? tau[i] : 0.0;
        a.inv_tau[i] = a.tau[i] > 0.0 ? 1.0 / a.tau[i] : 0.0;
        a.wgt[i] = i < n_den ? wgt[i] : 1.0;
    }
    double *d = nullptr, *nz = nullptr;
    WT_TRY(plan64_base(p, dst, &d));
    if (noise_plane != WT_PLANE_NONE) WT_TRY(plan64_base(p, noise_plane, &nz));
    ProfScope ps(p->ctx, "wt64_denoise_sum_kernel");
    hipLaunchKernelGGL(wt64_denoise_sum_kernel, dim3(flat_grid64(p)), dim3(256), 0, p->ctx->stream, a, (const double *)nz, d, plan64_n2(p));
    WT_HIP(hipGetLastError());
    return 0;
}

extern "C" int wt64_anscombe(wt_plan64 *p, int src, int dst, double alpha, double g, double sigma, int inverse)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_anscombe: null plan");
    if (alpha == 0.0) WT_FAIL("wt64_anscombe: alpha must be non-zero");
    double *s = nullptr, *d = nullptr;
    WT_TRY(plan64_base(p, src, &s));
    WT_TRY(plan64_base(p, dst, &d));
    hipLaunchKernelGGL(wt64_anscombe_kernel, grid64(p), dim3(256), 0, p->ctx->stream, (const double *)s, d, p->g.W, p->g.P, p->g.nrows, alpha, g, sigma, inverse);
    WT_HIP(hipGetLastError());
    return 0;
}

/* np.median(np.abs(plane)) in float64 (watroo/wavelets.py:127): exact.
 * Round 4: two radix levels of 11 bits (the first one rides on the transform's first fused pass when
 * flag bit4 asked for it), then ONE pass gathers the keys of the selected bin - a few ten thousand on
 * continuous data - and one workgroup finishes the select on that list, upper median of an even count
 * included: two or three reads of the plane instead of six or seven.  Bins that do not fit the list
 * (ties: constant or quantised data) continue with the radix passes as before. */
static int g_opt_select64_list = getenv("WT_NO_SELECT64_LIST") ? 0 : 1;      // wt_set_option("select64_list", 0/1): A/B and tests
void wt_set_select64_list(int on) { g_opt_select64_list = on; }
extern "C" int wt64_abs_median(wt_plan64 *p, int plane, double *median)
{
    WtGuard guard_(ctx_of(p));
    if (!p || !median) WT_FAIL("wt64_abs_median: null pointer");
    wt_ctx *c = p->ctx;
    const bool side = wt_wow_overlap_enabled() && p->overlap_ok && plane >= 0 && plane < p->overlap_scales;   // as wt_abs_median
    WtSideScope side_scope(c, side ? p->scale_ev[plane] : nullptr, side);
    if (!side_scope.ok()) return 2;
    const bool pre = c->prehist_plan == p && c->prehist_plane == plane;
    c->prehist_plan = nullptr;                                   // the bins are shared with the float32 select
    double *b = nullptr;
    WT_TRY(plan64_base(p, plane, &b));
    const int64_t N = (int64_t)p->g.nrows * p->g.W;
    const int64_t klo = (N - 1) / 2;
    Select64State *st = (Select64State *)(c->d_hist + WT_HIST_BINS + 16);
    Select64State *hst = (Select64State *)c->h_pinned;
    const bool windowed = pre && c->prehist_windowed;
    c->prehist_windowed = false;
    const int shifts[6] = {52, 41, 30, 19, 8, 0};
    const int bits[6] = {11, 11, 11, 11, 11, 8};
    // (work items are (row, chunk of 2048 samples) pairs: the grid is sized to the device, 8 blocks per CU)
    static const int hist_bpc = getenv("WT_HIST64_BLOCKS_PER_CU") ? std::max(1, atoi(getenv("WT_HIST64_BLOCKS_PER_CU"))) : 8;
    const int hgrid = (int)std::min<int64_t>((int64_t)p->g.nrows * ((p->g.W + 2047) / 2048), (int64_t)hist_bpc * c->num_cus);
    unsigned long long mask = 0;
    auto level = [&](int i, bool have_hist) -> int {
        const uint32_t bin_mask = (1u << bits[i]) - 1u;
        if (!have_hist) {
            ProfScope ps(c, "wt64_hist_kernel");
            hipLaunchKernelGGL(wt64_hist_kernel, dim3(hgrid), dim3(256), 0, c->stream, (const double *)b, p->g.nrows, p->g.P, p->g.W, mask,
                               (const Select64State *)st, shifts[i], bin_mask, c->d_hist);
        }
        hipLaunchKernelGGL(wt64_select_step_kernel, dim3(1), dim3(256), 0, c->stream, c->d_hist, st, (int)bin_mask + 1, shifts[i], i == 5);
        mask |= (unsigned long long)bin_mask << shifts[i];
        WT_HIP(hipGetLastError());
        return 0;
    };
    auto read_state = [&](Select64State *out) -> int {
        WT_HIP(hipMemcpyAsync((char *)c->h_pinned + 64, st, sizeof(Select64State), hipMemcpyDeviceToHost, c->stream));
        WT_HIP(hipStreamSynchronize(c->stream));
        *out = *(const Select64State *)((const char *)c->h_pinned + 64);
        return 0;
    };
    Select64State res{};
    // one attempt: the first 22 bits from the riding histogram (windowed: one step; plain: its step and a
    // second level over the plane) or from two passes, then the gathered list, else the radix passes
    auto attempt = [&](bool have_hist, bool window) -> int {
        memset(hst, 0, sizeof *hst);
        hst->k = (unsigned long long)klo;
        hst->upper = ~0ull;
        WT_HIP(hipMemcpyAsync(st, hst, sizeof(Select64State), hipMemcpyHostToDevice, c->stream));
        if (!have_hist) WT_HIP(hipMemsetAsync(c->d_hist, 0, WT_HIST_BINS * sizeof(uint32_t), c->stream));
        mask = 0;
        if (window) {
            hipLaunchKernelGGL(wt64_select_window_step_kernel, dim3(1), dim3(256), 0, c->stream, c->d_hist, st, (const uint32_t *)hist_base_word(c));
            WT_HIP(hipGetLastError());
            mask = 0x7ffffe0000000000ull;                    // the top 22 bits
        } else {
            WT_TRY(level(0, have_hist));
            WT_TRY(level(1, false));
        }
        bool listed = false;
        if (g_opt_select64_list) {
            const size_t cap = (size_t)1 << 20;
            if (!c->d_cand) {
                WT_HIP(hipMalloc(&c->d_cand, (cap + 1) * sizeof(unsigned long long)));
                c->d_cand_cap = cap;
            }
            WT_HIP(hipMemsetAsync(c->d_cand + c->d_cand_cap, 0, sizeof(unsigned long long), c->stream));
            {
                ProfScope ps(c, "wt64_collect_kernel");
                hipLaunchKernelGGL(wt64_collect_kernel, dim3(hgrid), dim3(256), 0, c->stream, (const double *)b, p->g.nrows, p->g.P, p->g.W, mask,
                                   (const Select64State *)st, c->d_cand, (unsigned long long)c->d_cand_cap);
            }
            hipLaunchKernelGGL(wt64_list_select_kernel, dim3(1), dim3(1024), 0, c->stream, (const unsigned long long *)c->d_cand,
                               (unsigned long long)c->d_cand_cap, st, 41);
            WT_HIP(hipGetLastError());
            WT_TRY(read_state(&res));
            if (res.failed == 3) return 0;                   // the window missed: the caller redoes the select
            listed = res.failed == 0;
            if (res.failed == 2) {                           // the bin did not fit the list: the state is untouched
                hst->failed = 0;
                WT_HIP(hipMemcpyAsync(&st->failed, &hst->failed, sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
            }
        }
        if (!listed) {
            if (res.failed == 1) WT_FAIL("wt64_abs_median: rank %lld not found (NaN input?)", (long long)klo);
            for (int i = 2; i < 6; ++i) WT_TRY(level(i, false));
            WT_TRY(read_state(&res));
            res.upper = ~0ull;
        }
        return 0;
    };
    WT_TRY(attempt(pre, windowed));
    if (windowed && res.failed == 3) WT_TRY(attempt(false, false));
    if (res.failed) WT_FAIL("wt64_abs_median: rank %lld not found (NaN input?)", (long long)klo);
    const unsigned long long ulo = res.prefix;
    unsigned long long uhi = ulo;
    if ((N & 1) == 0 && (int64_t)res.cum_le < klo + 2) {
        // the upper median is the smallest element above the lower one: among the gathered keys, or -
        // when the lower median is the largest of them / nothing was gathered - one more pass
        uhi = res.upper;
        if (uhi == ~0ull) {
            unsigned long long *r = (unsigned long long *)(c->d_hist + WT_HIST_BINS + 32);
            WT_HIP(hipMemsetAsync(r, 0xff, sizeof(unsigned long long), c->stream));
            {
                ProfScope ps(c, "wt64_min_greater_kernel");
                hipLaunchKernelGGL(wt64_min_greater_kernel, dim3(hgrid), dim3(256), 0, c->stream, (const double *)b, p->g.nrows, p->g.P,
                                   p->g.W, ulo, r);
            }
            WT_HIP(hipGetLastError());
            WT_HIP(hipMemcpyAsync(c->h_pinned, r, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
            WT_HIP(hipStreamSynchronize(c->stream));
            uhi = *(const unsigned long long *)c->h_pinned;
            if (uhi == ~0ull) WT_FAIL("wt64_abs_median: upper median not found");
        }
    }
    double lo, hi;
    memcpy(&lo, &ulo, 8);
    memcpy(&hi, &uhi, 8);
    *median = (N & 1) ? lo : (lo + hi) / 2.0;
    return 0;
}

/* one scale of the wow loop (watroo/utils.py:193-203): plane <- plane * significance(tau); gamma_plane
 * += plane (or WT_PLANE_NONE); plane <- plane * factor / sqrt(clip(power_plane, 1e-15)) (or
 * WT_PLANE_NONE: plane * factor) */
extern "C" int wt64_wow_update(wt_plan64 *p, int plane, int power_plane, double tau, int soft, int noise_plane, double factor,
                               int gamma_plane)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_wow_update: null plan");
    double *c = nullptr, *pw = nullptr, *nz = nullptr, *gm = nullptr;
    WT_TRY(plan64_base(p, plane, &c));
    if (power_plane != WT_PLANE_NONE) WT_TRY(plan64_base(p, power_plane, &pw));
    if (noise_plane != WT_PLANE_NONE) WT_TRY(plan64_base(p, noise_plane, &nz));
    if (gamma_plane != WT_PLANE_NONE) WT_TRY(plan64_base(p, gamma_plane, &gm));
    if (pw == c || gm == c || nz == c) WT_FAIL("wt64_wow_update: the plane aliases one of its operands");
    hipLaunchKernelGGL(wt64_wow_kernel, grid64(p), dim3(256), 0, p->ctx->stream, c, (const double *)pw, (const double *)nz, gm, p->g.W, p->g.P,
                       p->g.nrows, tau, soft, factor);
    WT_HIP(hipGetLastError());
    return 0;
}

/* One scale of the wow loop on an image (watroo/utils.py:193-203): local power conv_s(c^2) and the update of
 * wt64_wow_update in two kernels - the row pass of the squares into a private temporary, then the column
 * pass with the update as its epilogue, in place.  Same operations in the same order as wt64_smooth(square)
 * + wt64_wow_update: identical bits. */
extern "C" int wt64_wow_scale(wt_plan64 *p, int plane, int s, double tau, int soft, int noise_plane, double factor, int gamma_plane)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_wow_scale: null plan");
    if (s < 0 || s > 24) WT_FAIL("wt64_wow_scale: scale %d out of range", s);
    if (p->g.border != 0) WT_FAIL("wt64_wow_scale: images under the symmetric border only");
    // (as wt_wow_scale: behind a bilateral transform the update of w_s runs beside the later scales)
    const bool side = wt_wow_overlap_enabled() && p->overlap_ok && plane >= 0 && plane < p->overlap_scales &&
                      noise_plane == WT_PLANE_NONE && gamma_plane == WT_PLANE_NONE && stencil64_ok(p);
    WtSideScope side_scope(p->ctx, side ? p->scale_ev[plane] : nullptr, side);
    if (!side_scope.ok()) return 2;
    double *c = nullptr, *nz = nullptr, *gm = nullptr, *t1 = nullptr;
    WT_TRY(plan64_base(p, plane, &c));
    if (noise_plane != WT_PLANE_NONE) WT_TRY(plan64_base(p, noise_plane, &nz));
    if (gamma_plane != WT_PLANE_NONE) WT_TRY(plan64_base(p, gamma_plane, &gm));
    if (gm == c || nz == c) WT_FAIL("wt64_wow_scale: the plane aliases one of its operands");
    if (stencil64_ok(p) && plane >= 0 && plane <= p->max_level) {
        // built-in taps: local power, significance, gamma sum and whitening in ONE kernel (the fused wow update
        // of wt_stencil.h) writing a spare plane whose pointer is then swapped with the coefficient plane - as
        // wt_wow_scale does in float32
        double *spare = nullptr;
        WT_TRY(plan64_tmp(p, 0, &spare));
        ChainArgsT<double> a{};
        a.in = c; a.out_c = spare;
        a.noise = nz; a.gamma = gm; a.tau = tau; a.factor = factor; a.soft = soft; a.whiten = 1;
        WT_TRY(wt64_stencil_launch(stencil64_ctx(p), !nz && !gm ? MODE_WOW_PLAIN : (!nz ? MODE_WOW_GAMMA : MODE_WOW), a, s));
        std::swap(p->coef[plane], p->tmp[0]);
        return 0;
    }
    WT_TRY(plan64_tmp(p, 0, &t1));
    const Taps64 t = taps64(p);
    const int d = 1 << s;
    const bool even_w = g_opt_f64_pairs && p->g.W % 2 == 0;
    if (even_w && d % 2 == 0)
        hipLaunchKernelGGL(wt64_rows2_kernel, grid64_pairs(p), dim3(256), 0, p->ctx->stream, (const double *)c, t1, p->g, d, t, 1);
    else
        hipLaunchKernelGGL(wt64_rows_kernel, grid64(p), dim3(256), 0, p->ctx->stream, (const double *)c, t1, p->g, d, t, 1);
    if (even_w)
        hipLaunchKernelGGL(wt64_wow_axis2_kernel, grid64_pairs(p), dim3(256), 0, p->ctx->stream, (const double *)t1, c, (const double *)nz, gm,
                           p->g.W, p->g.P, p->g.nrows, d, p->g.border, t, tau, soft, factor);
    else
        hipLaunchKernelGGL(wt64_wow_axis_kernel, grid64(p), dim3(256), 0, p->ctx->stream, (const double *)t1, c, (const double *)nz, gm, p->g.W,
                           p->g.P, p->g.nrows, d, p->g.border, t, tau, soft, factor);
    WT_HIP(hipGetLastError());
    return 0;
}

/* gamma blend (watroo/utils.py:212-217): recon <- (1-h) recon + h clip((gamma-gmin)/(gmax-gmin))^inv_gamma */
extern "C" int wt64_gamma_blend(wt_plan64 *p, int recon, int gamma_plane, double gmin, double gmax, double inv_gamma, double h)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_gamma_blend: null plan");
    double *r = nullptr, *g = nullptr;
    WT_TRY(plan64_base(p, recon, &r));
    WT_TRY(plan64_base(p, gamma_plane, &g));
    hipLaunchKernelGGL(wt64_gamma_kernel, grid64(p), dim3(256), 0, p->ctx->stream, r, (const double *)g, p->g.W, p->g.P, p->g.nrows, gmin,
                       gmax - gmin, inv_gamma, h);
    WT_HIP(hipGetLastError());
    return 0;
}

extern "C" int wt64_fill_plane(wt_plan64 *p, int plane, double value)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_fill_plane: null plan");
    double *b = nullptr;
    WT_TRY(plan64_base(p, plane, &b));
    hipLaunchKernelGGL(wt64_fill_kernel, grid64(p), dim3(256), 0, p->ctx->stream, b, p->g.W, p->g.P, p->g.nrows, value);
    WT_HIP(hipGetLastError());
    return 0;
}

/* {sum, sum of squares, min, max} of a plane in float64 (c.std(), min / max of watroo/utils.py:176-214) */
extern "C" int wt64_reduce(wt_plan64 *p, int plane, double out[4])
{
    WtGuard guard_(ctx_of(p));
    if (!p || !out) WT_FAIL("wt64_reduce: null pointer");
    wt_ctx *c = p->ctx;
    double *b = nullptr;
    WT_TRY(plan64_base(p, plane, &b));
    const int blocks = std::min(p->g.nrows, c->partial_blocks);
    double *dout = c->d_partials + (size_t)c->partial_blocks * 4;
    {
        ProfScope ps(c, "wt64_reduce_kernel");
        hipLaunchKernelGGL(wt64_reduce_kernel, dim3(blocks), dim3(256), 0, c->stream, (const double *)b, p->g.nrows, p->g.P, p->g.W, c->d_partials);
        hipLaunchKernelGGL(wt_reduce_final_kernel, dim3(1), dim3(256), 0, c->stream, (const double *)c->d_partials, blocks, dout);
    }
    WT_HIP(hipGetLastError());
    WT_HIP(hipMemcpyAsync(c->h_pinned, dout, 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    WT_HIP(hipStreamSynchronize(c->stream));
    memcpy(out, c->h_pinned, 4 * sizeof(double));
    return 0;
}

/* atrous_convolution(image, kernel, bilateral_variance, s) (watroo/wavelets.py:74-105) in float64;
 * depth as for wt64_smooth; taps_reversed: the plan's taps are stored reversed (1-D signals) */
extern "C" int wt64_bilateral_conv(wt_plan64 *p, int src, int var, int dst, int s, int depth, int taps_reversed)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_bilateral_conv: null plan");
    if (s < 0 || s > 24) WT_FAIL("wt64_bilateral_conv: scale %d out of range", s);
    if (depth < 0 || (depth > 0 && p->g.H % depth)) WT_FAIL("wt64_bilateral_conv: height %d is not a multiple of depth %d", p->g.H, depth);
    if (p->g.border != 0 && p->g.border != 1) WT_FAIL("wt64_bilateral_conv: symmetric border (whole array or polyphase) only");
    if (src == dst || var == dst) WT_FAIL("wt64_bilateral_conv: dst must differ from src and var");
    double *in = nullptr, *v = nullptr, *o = nullptr;
    WT_TRY(plan64_base(p, src, &in));
    WT_TRY(plan64_base(p, var, &v));
    WT_TRY(plan64_base(p, dst, &o));
    if (depth == 0 && stencil64_ok(p))     // images, built-in (symmetric) taps: the marching kernel of wt_bilateral64.h
        return wt64_bilateral_launch(stencil64_ctx(p), in, v, o, nullptr, s, 1.0, 1.0);
    hipLaunchKernelGGL(wt64_bilateral_kernel, grid64(p), dim3(256), 0, p->ctx->stream, (const double *)in, (const double *)v, o, p->g,
                       depth > 0 ? p->g.H / depth : p->g.H, depth, 1 << s, taps64(p), taps_reversed);
    WT_HIP(hipGetLastError());
    return 0;
}

/* AtrousTransform(bilateral=...)(image, level) in float64 (watroo/wavelets.py:408-444 with :433-440): per scale
 * the variance of sdev_loc times sigma_b[s]**2 (times s + 1 under bilateral_scaling), the range-weighted filter
 * and the detail plane c_s - c_{s+1}; planes 0 .. level.  As wt_decompose_bilateral in float32: with built-in
 * taps one marching kernel per scale forms the variance in its register window and writes both planes; other
 * taps take the three generic kernels per scale. */
extern "C" int wt64_decompose_bilateral(wt_plan64 *p, int src, int level, const double *sigma_b, int bilateral_scaling)
{
    WtGuard guard_(ctx_of(p));
    if (!p || !sigma_b) WT_FAIL("wt64_decompose_bilateral: null pointer");
    if (level < 0 || level > p->max_level) WT_FAIL("wt64_decompose_bilateral: level %d exceeds plan max_level %d", level, p->max_level);
    if (src >= 0 && src <= level) WT_FAIL("wt64_decompose_bilateral: src plane %d is one of the output planes", src);
    if (src <= WT_PLANE_SCRATCH(0) && src >= WT_PLANE_SCRATCH(1)) WT_FAIL("wt64_decompose_bilateral: scratch planes 0 and 1 are used internally");
    if (p->g.border != 0) WT_FAIL("wt64_decompose_bilateral: images under the symmetric border only");
    double *in = nullptr;
    WT_TRY(plan64_base(p, src, &in));
    if (level == 0) {
        double *o = nullptr;
        WT_TRY(plan64_base(p, 0, &o));
        WT_HIP(hipMemcpyAsync(o, in, (size_t)p->g.nrows * p->g.P * sizeof(double), hipMemcpyDeviceToDevice, p->ctx->stream));
        return 0;
    }
    const bool tiled = stencil64_ok(p);
    const bool overlap = wt_wow_overlap_enabled();
    if (overlap) WT_TRY(wt_scale_events(p->ctx, p->scale_ev, level));
    int cur = src;
    for (int s = 0; s < level; ++s) {
        if (s > 24) WT_FAIL("wt64_decompose_bilateral: scale %d out of range", s);
        const int nxt = (s == level - 1) ? level : WT_PLANE_SCRATCH(s & 1);
        double *oc = nullptr, *ow = nullptr;
        WT_TRY(plan64_base(p, cur, &in));
        WT_TRY(plan64_base(p, nxt, &oc));
        WT_TRY(plan64_base(p, s, &ow));
        const double f1 = sigma_b[s] * sigma_b[s];                       // watroo/wavelets.py:434-436
        const double f2 = bilateral_scaling ? (double)(s + 1) : 1.0;
        if (tiled) {
            WT_TRY(wt64_bilateral_launch(stencil64_ctx(p), in, nullptr, oc, ow, s, f1, f2));
        } else {
            double *var = nullptr, *mean = nullptr;
            WT_TRY(plan64_tmp(p, 1, &var));
            WT_TRY(plan64_tmp(p, 2, &mean));
            WT_TRY(smooth64(p, in, mean, nullptr, s, 0, 0));
            WT_TRY(smooth64(p, in, var, nullptr, s, 1, 0));
            hipLaunchKernelGGL(wt64_var_kernel, grid64(p), dim3(256), 0, p->ctx->stream, (const double *)mean, (const double *)var, var, p->g.W,
                               p->g.P, p->g.nrows, f1, f2, 0);
            hipLaunchKernelGGL(wt64_bilateral_kernel, grid64(p), dim3(256), 0, p->ctx->stream, (const double *)in, (const double *)var, oc, p->g,
                               p->g.H, 0, 1 << s, taps64(p), 0);
            hipLaunchKernelGGL(wt64_binary_kernel, grid64(p), dim3(256), 0, p->ctx->stream, (const double *)in, (const double *)oc, ow, p->g.W, p->g.P,
                               p->g.nrows, 1);
            WT_HIP(hipGetLastError());
        }
        if (overlap) WT_HIP(hipEventRecord(p->scale_ev[s], p->ctx->stream));     // w_s is written
        cur = nxt;
    }
    // the per-scale work on w_s that follows (wt64_wow_scale, wt64_abs_median) may run beside the scales still queued
    p->overlap_scales = overlap ? level : 0;
    p->overlap_ok = overlap;
    return 0;
}

/* wt_taps_conv in float64 */
extern "C" int wt64_taps_conv_ex(wt_plan64 *p, int src, int var, int dst, const int32_t *offsets, const double *weights, int ntaps,
                                 double center_weight, int has_center, int depth, int pad_mode, double fill_value, int dilation);
extern "C" int wt64_taps_conv(wt_plan64 *p, int src, int var, int dst, const int32_t *offsets, const double *weights, int ntaps,
                              double center_weight, int has_center, int depth, int pad_mode, double fill_value)
{
    WtGuard guard_(ctx_of(p));
    if (pad_mode > WT_PAD_CONSTANT) WT_FAIL("wt64_taps_conv: unknown pad mode %d (the polyphase modes take a dilation: wt64_taps_conv_ex)", pad_mode);
    return wt64_taps_conv_ex(p, src, var, dst, offsets, weights, ntaps, center_weight, has_center, depth, pad_mode, fill_value, 1);
}

/* wt_axis_filter in float64 */
extern "C" int wt64_axis_filter(wt_plan64 *p, int src, int dst, int axis, const int32_t *offsets, const double *weights, int ntaps, int depth,
                                int pad_mode, double fill_value, int dilation)
{
    WtGuard guard_(ctx_of(p));
    if (!p || !offsets || !weights) WT_FAIL("wt64_axis_filter: null pointer");
    if (ntaps < 1 || ntaps > 4096) WT_FAIL("wt64_axis_filter: %d taps unsupported", ntaps);
    if (axis < 0 || axis > 2) WT_FAIL("wt64_axis_filter: axis %d (2 = x, 1 = y, 0 = z)", axis);
    if (dilation < 1) WT_FAIL("wt64_axis_filter: dilation %d must be positive", dilation);
    if (src == dst) WT_FAIL("wt64_axis_filter: dst must differ from src");
    if (pad_mode < WT_PAD_SYMMETRIC || pad_mode > WT_PAD_POLY_MIRROR) WT_FAIL("wt64_axis_filter: unknown pad mode %d", pad_mode);
    if (depth < 0 || (depth > 0 && p->g.H % depth)) WT_FAIL("wt64_axis_filter: %d rows are not a multiple of depth %d", p->g.H, depth);
    if (axis == 0 && depth == 0) WT_FAIL("wt64_axis_filter: axis 0 needs a cube (depth > 0)");
    double *in = nullptr, *o = nullptr;
    WT_TRY(plan64_base(p, src, &in));
    WT_TRY(plan64_base(p, dst, &o));
    const int rc = wt_axis_filter_launch<double>(p->ctx, in, o, p->g.W, p->g.P, p->g.H, depth, axis, offsets, weights, ntaps, pad_mode,
                                                 fill_value, dilation);
    if (rc >= 0) return rc;
    std::vector<int32_t> o3((size_t)ntaps * 3, 0);
    for (int j = 0; j < ntaps; ++j) o3[(size_t)3 * j + axis] = offsets[j];
    return wt64_taps_conv_ex(p, src, WT_PLANE_NONE, dst, o3.data(), weights, ntaps, 0.0, 0, depth, pad_mode, fill_value, dilation);
}

extern "C" int wt64_variance_from_moments(wt_plan64 *p, int mean, int meansq, int dst, double f1, double f2, int take_sqrt)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_variance_from_moments: null plan");
    double *m = nullptr, *q = nullptr, *d = nullptr;
    WT_TRY(plan64_base(p, mean, &m));
    WT_TRY(plan64_base(p, meansq, &q));
    WT_TRY(plan64_base(p, dst, &d));
    hipLaunchKernelGGL(wt64_var_kernel, grid64(p), dim3(256), 0, p->ctx->stream, (const double *)m, (const double *)q, d, p->g.W, p->g.P, p->g.nrows, f1, f2,
                       take_sqrt);
    WT_HIP(hipGetLastError());
    return 0;
}

extern "C" int wt64_taps_conv_ex(wt_plan64 *p, int src, int var, int dst, const int32_t *offsets, const double *weights, int ntaps,
                                 double center_weight, int has_center, int depth, int pad_mode, double fill_value, int dilation)
{
    WtGuard guard_(ctx_of(p));
    if (!p || (ntaps > 0 && (!offsets || !weights))) WT_FAIL("wt64_taps_conv: null pointer");
    if (dilation < 1) WT_FAIL("wt64_taps_conv: dilation %d must be positive", dilation);
    if (src == dst || var == dst) WT_FAIL("wt64_taps_conv: dst must differ from src and var");
    if (pad_mode < WT_PAD_SYMMETRIC || pad_mode > WT_PAD_POLY_MIRROR) WT_FAIL("wt64_taps_conv: unknown pad mode %d", pad_mode);
    if (depth < 0 || (depth > 0 && p->g.H % depth)) WT_FAIL("wt64_taps_conv: %d rows are not a multiple of depth %d", p->g.H, depth);
    double *in = nullptr, *o = nullptr, *v = nullptr;
    WT_TRY(plan64_base(p, src, &in));
    WT_TRY(plan64_base(p, dst, &o));
    if (var != WT_PLANE_NONE) WT_TRY(plan64_base(p, var, &v));
    const int32_t *d_offs = nullptr;
    const double *d_wts = nullptr;
    WT_TRY(upload_taplist<double>(p->ctx, offsets, weights, ntaps, &d_offs, &d_wts));
    const int Z = depth > 0 ? depth : 1, Y = p->g.H / Z;
    hipLaunchKernelGGL(wt_taps_kernel<double>, grid64(p), dim3(256), 0, p->ctx->stream, (const double *)in, (const double *)v, o, p->g.W, p->g.P, Y, Z,
                       d_offs, d_wts, ntaps, center_weight, has_center, pad_mode, fill_value, dilation);
    WT_HIP(hipGetLastError());
    return 0;
}

/* dst[dst_plane][dy:dy+rows, dx:dx+cols] = src[src_plane][sy:sy+rows, sx:sx+cols] on the device (the
 * crop of atrous_recursive, watroo/wavelets.py:405-406) */
extern "C" int wt64_copy_window(wt_plan64 *src, int src_plane, wt_plan64 *dst, int dst_plane, int64_t sy, int64_t sx, int64_t dy,
                                int64_t dx, int64_t rows, int64_t cols)
{
    WtGuard guard_(ctx_of(src), ctx_of(dst));
    if (!src || !dst) WT_FAIL("wt64_copy_window: null plan");
    if (src->ctx != dst->ctx) WT_FAIL("wt64_copy_window: plans of different contexts");
    if (rows < 1 || cols < 1 || sy < 0 || sx < 0 || dy < 0 || dx < 0 || sy + rows > src->g.nrows || sx + cols > src->g.W ||
        dy + rows > dst->g.nrows || dx + cols > dst->g.W)
        WT_FAIL("wt64_copy_window: window outside a plane");
    double *a = nullptr, *b = nullptr;
    WT_TRY(plan64_base(src, src_plane, &a));
    WT_TRY(plan64_base(dst, dst_plane, &b));
    WT_HIP(hipMemcpy2DAsync(b + dy * dst->g.P + dx, (size_t)dst->g.P * 8, a + sy * src->g.P + sx, (size_t)src->g.P * 8, (size_t)cols * 8, (size_t)rows,
                            hipMemcpyDeviceToDevice, src->ctx->stream));
    return 0;
}

/* cv2.filter2D with a small kernel (host pointer, kh * kw doubles), explicit anchor, symmetric
 * (border 0) or periodic (border 3, WT_BORDER_PERIODIC) border: as wt_filter2d_ex, in float64 */
extern "C" int wt64_filter2d(wt_plan64 *p, int src, int dst, const double *kernel, int kh, int kw, int ay, int ax, int border)
{
    WtGuard guard_(ctx_of(p));
    if (!p || !kernel) WT_FAIL("wt64_filter2d: null pointer");
    if (kh < 1 || kw < 1 || (int64_t)kh * kw > (1 << 20)) WT_FAIL("wt64_filter2d: kernel %d x %d unsupported", kh, kw);
    if (ay < 0 || ay >= kh || ax < 0 || ax >= kw) WT_FAIL("wt64_filter2d: anchor (%d, %d) outside the %d x %d kernel", ay, ax, kh, kw);
    if (border != WT_BORDER_SYMMETRIC && border != WT_BORDER_PERIODIC) WT_FAIL("wt64_filter2d: border %d unsupported", border);
    if (src == dst) WT_FAIL("wt64_filter2d: src and dst must differ");
    double *in = nullptr, *o = nullptr;
    WT_TRY(plan64_base(p, src, &in));
    WT_TRY(plan64_base(p, dst, &o));
    const size_t n = (size_t)kh * kw;
    if (p->psf_cap < n) {
        void *q = nullptr;
        WT_HIP(hipMalloc(&q, n * sizeof(double)));
        p->allocs.push_back(q);
        p->psf = (double *)q;
        p->psf_cap = n;
    }
    // caller-owned pageable memory: drain the stream (the previous PSF may still be in use) and copy
    // synchronously - see upload_taplist
    WT_HIP(hipStreamSynchronize(p->ctx->stream));
    WT_HIP(hipMemcpy(p->psf, kernel, n * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(wt64_filter2d_kernel, grid64(p), dim3(256), 0, p->ctx->stream, (const double *)in, o, p->g, (const double *)p->psf, kh, kw,
                       ay, ax, border == WT_BORDER_PERIODIC);
    WT_HIP(hipGetLastError());
    return 0;
}

/* wt_fft_spectrum / wt_fft_apply in float64 */
extern "C" int wt64_fft_spectrum(wt_plan64 *p, int src)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_fft_spectrum: null plan");
    if (!wt_fft_size_ok(p->g.H, p->g.W)) WT_FAIL("wt64_fft_spectrum: a side of the %d x %d image has a prime factor above 5 (or lies outside 2 .. %d)", p->g.H, p->g.W, WT_FFT_MAX_N);
    double *s = nullptr;
    WT_TRY(plan64_base(p, src, &s));
    WT_TRY(wt_fft_prepare<double>(p->ctx, p->fft, p->g.H, p->g.W, p->allocs));
    return wt_fft_set_spectrum<double>(p->ctx, p->fft, s, p->g.P);
}

extern "C" int wt64_fft_apply(wt_plan64 *p, int src, int dst, int conj)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_fft_apply: null plan");
    if (!wt_fft_size_ok(p->g.H, p->g.W)) WT_FAIL("wt64_fft_apply: a side of the %d x %d image has a prime factor above 5 (or lies outside 2 .. %d)", p->g.H, p->g.W, WT_FFT_MAX_N);
    double *s = nullptr, *d = nullptr;
    WT_TRY(plan64_base(p, src, &s));
    WT_TRY(plan64_base(p, dst, &d));
    return wt_fft_apply_t<double>(p->ctx, p->fft, s, d, p->g.P, conj);
}

/* multiresolution-support update (watroo/utils.py:263-276), as wt_mrs_update */
extern "C" int wt64_mrs_update(wt_plan64 *p, int plane, int mrs_plane, double tau, int soft, int noise_plane, int persistent, double inv_pow)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt64_mrs_update: null plan");
    double *c = nullptr, *m = nullptr, *nz = nullptr;
    WT_TRY(plan64_base(p, plane, &c));
    WT_TRY(plan64_base(p, mrs_plane, &m));
    if (noise_plane != WT_PLANE_NONE) WT_TRY(plan64_base(p, noise_plane, &nz));
    if (c == m) WT_FAIL("wt64_mrs_update: plane and support plane must differ");
    hipLaunchKernelGGL(wt64_mrs_kernel, grid64(p), dim3(256), 0, p->ctx->stream, c, m, (const double *)nz, p->g.W, p->g.P, p->g.nrows, tau, soft,
                       persistent, inv_pow);
    WT_HIP(hipGetLastError());
    return 0;
}
