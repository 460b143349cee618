// Host-side internals shared by the translation units of libwatroo_hip.so (round 5: wt_api.hip was one file of
// 2 900 lines; it is now wt_core.hip - errors, profiling, RCCL, side stream, context, plans and their memory,
// host transfers, halo exchange - wt_transform.hip - per-scale launches and the decomposition drivers -
// wt_apps.hip - pointwise operators, cubes, richardson_lucy support, median and reductions - and wt_f64.hip, the
// float64 engine).  Nothing here is part of the C ABI (include/watroo_hip.h).
#pragma once
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "wt_internal.h"
#include "wt_stencil_launch.h"

// Per-context serialisation of the entry points (see wt_ctx::mu).  Two-plan operations lock both
// contexts in address order.  The guard also makes the (first) context's device the calling thread's
// current one: HIP's current device is per thread, so a host thread other than the one that created
// the context - or one that has since used a context on another GPU - would otherwise launch on a
// stream of a device that is not current.
void wt_ctx_warm_join(wt_ctx *c);      // (wt_core.hip; with the context's lock held)
struct WtGuard {
    std::recursive_mutex *a = nullptr, *b = nullptr;
    explicit WtGuard(wt_ctx *c, wt_ctx *d = nullptr)
    {
        const int dev = c ? c->device : -1;
        if (c == d) d = nullptr;
        if (c && d && d < c) std::swap(c, d);
        if (c) { a = &c->mu; a->lock(); if (c->warm) wt_ctx_warm_join(c); }
        if (d) { b = &d->mu; b->lock(); if (d->warm) wt_ctx_warm_join(d); }
        if (dev >= 0) (void)hipSetDevice(dev);
    }
    ~WtGuard()
    {
        if (b) b->unlock();
        if (a) a->unlock();
    }
    WtGuard(const WtGuard &) = delete;
    WtGuard &operator=(const WtGuard &) = delete;
};
static inline wt_ctx *ctx_of(wt_plan *p) { return p ? p->ctx : nullptr; }
static inline wt_ctx *ctx_of(wt_ctx *c) { return c; }

// ------------------------------------------------------------------ RCCL, loaded on demand (wt_core.hip)
struct RcclApi {
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    void *CommInitRank = nullptr;  // ncclCommInitRank(comm*, nranks, ncclUniqueId by value, rank)
    int (*CommDestroy)(void *) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    int (*CommUserRank)(void *, int *) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    int (*GetVersion)(int *) = nullptr;
};
struct UniqueId128 {
    char b[128];
};
typedef int (*CommInitRank_t)(void **, int, UniqueId128, int);

extern RcclApi g_rccl;
enum { NCCL_UINT32 = 3, NCCL_UINT64 = 5, NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8 };
enum { NCCL_SUM = 0, NCCL_MAX = 2, NCCL_MIN = 3 };

int rccl_load();
#define WT_NCCL(expr)                                                                         \
    do {                                                                                      \
        int r_ = (expr);                                                                      \
        if (r_ != 0) {                                                                        \
            wt_set_error("%s: RCCL error %d (%s) at %s:%d: %s", __func__, r_,                 \
                         g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?", __FILE__,   \
                         __LINE__, #expr);                                                    \
            return 3;                                                                         \
        }                                                                                     \
    } while (0)


// ------------------------------------------------------------------ options (wt_set_option, wt_transform.hip)
extern int g_opt_tri4, g_opt_scatter, g_opt_scatter_strips;                  // wt_core.hip
extern int g_opt_hist_window, g_opt_bilateral_paired, g_opt_overlap, g_opt_overlap_reserve, g_opt_split_dry, g_opt_host_pipeline;   // wt_transform.hip
extern int g_opt_wow_overlap;                                                // wt_core.hip
extern int g_opt_axis_filter;                                                // wt_apps.hip
void wt_set_fused64(int on);                                                 // wt_f64.hip
void wt_set_select64_list(int on);
void wt_set_f64_pairs(int on);
void wt_set_stencil64(int on);

// ------------------------------------------------------------------ plans (wt_core.hip)
static inline int family_taps(int family) { return family == WT_B3SPLINE ? 5 : 3; }
static inline int64_t plan_n4(const wt_plan *p) { return (int64_t)p->g.nrows * p->g.P / 4; }
static inline int flat_grid(int64_t n4) { return (int)std::min<int64_t>((n4 + 255) / 256, 256 * 8); }
int plane_base(wt_plan *p, int id, float **base);          // pointer to LOCAL ROW 0 of a plane (allocated on first use)
bool is_vmm(const wt_plan *p, const float *b);
int vmm_stage(wt_plan *p, float **stage);
int vmm_copy(wt_plan *p, float *dst, const float *src);
int copy2d(wt_plan *a, wt_plan *b, float *dst, size_t dpitch, const float *src, size_t spitch, size_t cols, size_t rows, hipStream_t st);
void destroy_events(std::vector<hipEvent_t> &ev);
bool try_pin(const void *host, size_t bytes);
int halo_exchange_on(wt_plan *p, int plane, int64_t rows, hipStream_t st, const char *prof_name = "rccl_halo_exchange");

// ------------------------------------------------------------------ per-scale launches (wt_transform.hip)
int check_scale(const wt_plan *p, int s, const char *who);
static inline StencilCtx stencil_ctx(const wt_plan *p, hipStream_t st = nullptr)
{
    return StencilCtx{p->ctx, st ? st : p->ctx->stream, p->g, p->family};
}
static inline int64_t scale_halo(const wt_plan *p, int s) { return (int64_t)(family_taps(p->family) / 2) << s; }
int maybe_exchange(wt_plan *p, int plane, int64_t rows, int flags);
// one scale in `mode` (MODE_* of wt_stencil.h): user-defined taps on the generic separable kernels, the built-in
// families on wt_stencil.h's (the units other than wt_transform.hip launch through this, not the templates)
int launch_chain_mode(wt_plan *p, int mode, ChainArgs a, int s, const char *name);
int launch_custom(wt_plan *p, const float *in, float *out_c, float *out_w, int s, int square, const char *name);
int launch_custom_variance(wt_plan *p, const float *in, float *out, int s, float f1, float f2, int take_sqrt, const char *name);
static inline uint32_t *hist_base_word(wt_ctx *c) { return c->d_hist + WT_HIST_BINS + 30; }

// ------------------------------------------------------------------ applications (wt_apps.hip)
int denoise_sum_rows(wt_plan *p, int count, int dst, int n_den, const double *tau, const double *wgt, int soft, int r0, int r1);

// Tap list of the generic operator on the device: [ntaps x 3 int32 offsets][ntaps x weight]; the
// previous list may still be in use by a kernel on the stream - the copy is stream-ordered.
#define WT_MAX_TAPLIST 65536
template <typename T>
static int upload_taplist(wt_ctx *c, const int32_t *offs, const T *wts, int ntaps, const int32_t **d_offs, const T **d_wts)
{
    if (ntaps < 0 || ntaps > WT_MAX_TAPLIST) WT_FAIL("tap list of %d entries (0..%d supported)", ntaps, WT_MAX_TAPLIST);
    const size_t need = (size_t)std::max(ntaps, 1);
    if (c->d_taps_cap < need) {
        WT_HIP(hipStreamSynchronize(c->stream));
        if (c->d_taps) (void)hipFree(c->d_taps);
        c->d_taps = nullptr;
        c->d_taps_cap = 0;
        WT_HIP(hipMalloc(&c->d_taps, need * (3 * sizeof(int32_t) + sizeof(double))));
        c->d_taps_cap = need;
    }
    // layout: [cap x 8-byte weight slots][cap x 3 int32 offsets] - the weights first, so that double
    // weights are 8-byte aligned whatever the capacity (hipMalloc returns 256-byte aligned blocks)
    char *base = (char *)c->d_taps;
    char *offs_base = base + c->d_taps_cap * sizeof(double);
    // The list comes from caller-owned pageable memory that may be freed the moment this call returns
    // (temporaries of the Python layer), and the previous list may still be read by a kernel on the
    // stream: drain the stream, then copy SYNCHRONOUSLY.  (An asynchronous copy from such memory is a
    // use-after-free in waiting: a GPU memory fault on a host heap address showed up once in a long
    // fuzz run.)  The generic operator is a correctness path; the drain costs microseconds.
    if (ntaps) {
        WT_HIP(hipStreamSynchronize(c->stream));
        WT_HIP(hipMemcpy(offs_base, offs, (size_t)ntaps * 3 * sizeof(int32_t), hipMemcpyHostToDevice));
        WT_HIP(hipMemcpy(base, wts, (size_t)ntaps * sizeof(T), hipMemcpyHostToDevice));
    }
    *d_offs = (const int32_t *)offs_base;
    *d_wts = (const T *)base;
    return 0;
}

