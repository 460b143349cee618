// libwatroo_hip.so - host side of the C ABI declared in include/watroo_hip.h.
// Plan / buffer management, kernel dispatch, RCCL halo exchange.  gfx950 only.
#include <dlfcn.h>
#include <sys/mman.h>

#include <algorithm>
#include <thread>
#include <cstdarg>
#include <cstdlib>

#include "wt_internal.h"
#include "wt_kernels.h"
#include "wt_stencil_launch.h"
#include "wt_fused_decl.h"
#include "wt_fft.h"
#include "wt_rccl_group.h"
#include "wt_axis.h"

// =============================================================================================
// errors
// =============================================================================================
static thread_local char g_err[512] = "";

void wt_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *wt_last_error(void) { return g_err; }

// Per-context serialisation of the entry points (see wt_ctx::mu).  Two-plan operations lock both
// contexts in address order.  The guard also makes the (first) context's device the calling thread's
// current one: HIP's current device is per thread, so a host thread other than the one that created
// the context - or one that has since used a context on another GPU - would otherwise launch on a
// stream of a device that is not current.
struct WtGuard {
    std::recursive_mutex *a = nullptr, *b = nullptr;
    explicit WtGuard(wt_ctx *c, wt_ctx *d = nullptr)
    {
        const int dev = c ? c->device : -1;
        if (c == d) d = nullptr;
        if (c && d && d < c) std::swap(c, d);
        if (c) { a = &c->mu; a->lock(); }
        if (d) { b = &d->mu; b->lock(); }
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
extern "C" int wt_abi_version(void) { return WT_ABI_VERSION; }

extern "C" int wt_device_count(int *count)
{
    if (!count) WT_FAIL("wt_device_count: null pointer");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        wt_set_error("hipGetDeviceCount failed: %d (%s)", (int)e, hipGetErrorString(e));
        (void)hipGetLastError();
        n = 0;
    } else {
        wt_set_error("hipGetDeviceCount: %d device(s)", n);
    }
    *count = n;
    return 0;
}

// =============================================================================================
// profiling scope
// =============================================================================================
static hipEvent_t take_event(wt_ctx *c)
{
    if (!c->event_pool.empty()) {
        hipEvent_t e = c->event_pool.back();
        c->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

ProfScope::ProfScope(wt_ctx *c, const char *n, hipStream_t s) : ctx(c), name(n), st(s ? s : c->stream)
{
    if (!ctx->profiling) return;
    a = take_event(ctx);
    b = take_event(ctx);
    (void)hipEventRecord(a, st);
}

ProfScope::~ProfScope()
{
    if (!ctx->profiling || !a) return;
    (void)hipEventRecord(b, st);
    ctx->pending.push_back({name, a, b});
}

static int prof_resolve(wt_ctx *c)
{
    if (c->pending.empty()) return 0;
    WT_HIP(hipStreamSynchronize(c->stream));
    if (c->comm_stream) WT_HIP(hipStreamSynchronize(c->comm_stream));
    if (c->side_stream) WT_HIP(hipStreamSynchronize(c->side_stream));
    for (auto &p : c->pending) {
        float ms = 0.f;
        WT_HIP(hipEventElapsedTime(&ms, p.a, p.b));
        auto it = c->prof.find(p.name);
        if (it == c->prof.end()) {
            c->prof_order.push_back(p.name);
            it = c->prof.emplace(p.name, ProfEntry{}).first;
        }
        it->second.calls += 1;
        it->second.ms += ms;
        c->event_pool.push_back(p.a);
        c->event_pool.push_back(p.b);
    }
    c->pending.clear();
    return 0;
}

// =============================================================================================
// RCCL, loaded on demand
// =============================================================================================
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

static RcclApi g_rccl;
enum { NCCL_UINT32 = 3, NCCL_UINT64 = 5, NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8 };
enum { NCCL_SUM = 0, NCCL_MAX = 2, NCCL_MIN = 3 };

static int rccl_load()
{
    static std::mutex load_mu;
    std::lock_guard<std::mutex> lk(load_mu);
    if (g_rccl.h) return 0;
    // WATROO_HIP_RCCL_LIB: another library with RCCL's entry points (tests load a stub whose ncclSend fails on
    // demand, tests/stubs/rccl_stub.c)
    const char *over = getenv("WATROO_HIP_RCCL_LIB");
    void *h = over && *over ? dlopen(over, RTLD_NOW | RTLD_GLOBAL) : nullptr;
    if (over && *over && !h) WT_FAIL("cannot load WATROO_HIP_RCCL_LIB=%s: %s", over, dlerror());
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) WT_FAIL("cannot load librccl.so: %s", dlerror());
#define SYM(field, name)                                              \
    *(void **)(&g_rccl.field) = dlsym(h, name);                      \
    if (!g_rccl.field) WT_FAIL("librccl.so lacks symbol %s", name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(CommCount, "ncclCommCount");
    SYM(CommUserRank, "ncclCommUserRank");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(AllReduce, "ncclAllReduce");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    *(void **)(&g_rccl.GetVersion) = dlsym(h, "ncclGetVersion");      // (optional)
    g_rccl.h = h;
    return 0;
}

#define WT_NCCL(expr)                                                                         \
    do {                                                                                      \
        int r_ = (expr);                                                                      \
        if (r_ != 0) {                                                                        \
            wt_set_error("RCCL error %d (%s) at %s:%d: %s", r_,                               \
                         g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?", __FILE__,   \
                         __LINE__, #expr);                                                    \
            return 3;                                                                         \
        }                                                                                     \
    } while (0)

extern "C" int wt_comm_unique_id(void *id128)
{
    if (!id128) WT_FAIL("wt_comm_unique_id: null pointer");
    WT_TRY(rccl_load());
    WT_NCCL(g_rccl.GetUniqueId(id128));
    return 0;
}

extern "C" int wt_ctx_comm_init(wt_ctx *ctx, int rank, int nranks, const void *id128)
{
    WtGuard guard_(ctx_of(ctx));
    if (!ctx || !id128) WT_FAIL("wt_ctx_comm_init: null pointer");
    if (nranks < 1 || rank < 0 || rank >= nranks) WT_FAIL("wt_ctx_comm_init: bad rank %d/%d", rank, nranks);
    if (ctx->comm) WT_FAIL("wt_ctx_comm_init: communicator already initialised");
    WT_TRY(rccl_load());
    WT_HIP(hipSetDevice(ctx->device));
    UniqueId128 id;
    memcpy(&id, id128, sizeof(id));
    void *comm = nullptr;
    WT_NCCL(((CommInitRank_t)g_rccl.CommInitRank)(&comm, nranks, id, rank));
    ctx->comm = comm;
    ctx->rank = rank;
    ctx->nranks = nranks;
    if (!ctx->comm_stream) {
        int lo = 0, hi = 0;   // numerically lowest value = highest priority
        WT_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        WT_HIP(hipStreamCreateWithPriority(&ctx->comm_stream, hipStreamNonBlocking, hi));
        WT_HIP(hipEventCreateWithFlags(&ctx->ev_to_comm, hipEventDisableTiming));
        WT_HIP(hipEventCreateWithFlags(&ctx->ev_from_comm, hipEventDisableTiming));
    }
    return 0;
}

// What the COMMUNICATOR says about itself (ncclCommCount / ncclCommUserRank), not what the caller
// passed to wt_ctx_comm_init: bench.py reports it as `rccl_ranks`.  No communicator: 0 / 1.
extern "C" int wt_ctx_comm_info(wt_ctx *ctx, int *rank, int *nranks)
{
    WtGuard guard_(ctx_of(ctx));
    if (!ctx || !rank || !nranks) WT_FAIL("wt_ctx_comm_info: null pointer");
    *rank = 0;
    *nranks = 1;
    if (!ctx->comm) return 0;
    WT_NCCL(g_rccl.CommCount(ctx->comm, nranks));
    WT_NCCL(g_rccl.CommUserRank(ctx->comm, rank));
    return 0;
}

// What the LIBRARY says its version is (ncclGetVersion: major * 10000 + minor * 100 + patch for 2.9 and later) -
// bench.py prints it into the multi-GPU line next to `rccl_ranks`.  0: the symbol is missing.
extern "C" int wt_comm_version(int *version)
{
    if (!version) WT_FAIL("wt_comm_version: null pointer");
    *version = 0;
    WT_TRY(rccl_load());
    if (g_rccl.GetVersion) WT_NCCL(g_rccl.GetVersion(version));
    return 0;
}

// "device=<hip ordinal> pci=<domain:bus:device.function> cus=<n> name=<marketing name>" of the context's GPU:
// the multi-GPU bench line lists it per rank, so that a run on a shared or mis-bound node explains itself.
extern "C" int wt_ctx_device_info(wt_ctx *c, char *buf, int cap)
{
    WtGuard guard_(ctx_of(c));
    if (!c || !buf || cap < 16) WT_FAIL("wt_ctx_device_info: null pointer or a buffer below 16 bytes");
    char pci[64] = "?";
    if (hipDeviceGetPCIBusId(pci, (int)sizeof pci, c->device) != hipSuccess) {
        (void)hipGetLastError();
        snprintf(pci, sizeof pci, "?");
    }
    hipDeviceProp_t prop{};
    const char *name = "?";
    if (hipGetDeviceProperties(&prop, c->device) == hipSuccess) name = prop.name[0] ? prop.name : prop.gcnArchName;   // (containers often lack the marketing name)
    else (void)hipGetLastError();
    snprintf(buf, (size_t)cap, "device=%d pci=%s cus=%d name=%s", c->device, pci, c->num_cus, name);
    return 0;
}

// =============================================================================================
// side stream
// =============================================================================================
extern int g_opt_axis_filter;
static int g_opt_wow_overlap = getenv("WT_NO_WOW_OVERLAP") ? 0 : 1;   // wt_set_option("wow_overlap", 0/1)
bool wt_wow_overlap_enabled() { return g_opt_wow_overlap != 0; }

int wt_side_join(wt_ctx *c)
{
    if (!c->side_pending || c->in_side) return 0;
    WT_HIP(hipEventRecord(c->ev_side_done, c->side_stream));
    WT_HIP(hipStreamWaitEvent(c->stream, c->ev_side_done, 0));
    c->side_pending = false;
    return 0;
}

int wt_side_begin(wt_ctx *c, hipEvent_t after)
{
    if (c->in_side) WT_FAIL("side stream: nested use");
    if (!c->side_stream) {
        int lo = 0, hi = 0;
        WT_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
        // default priority: measured on cfg5 (8192^2, tools/ab_overlap.sh) 6.375 ms against 6.474 with the high
        // priority and 6.506 without the side stream
        static const int prio_env = getenv("WT_SIDE_PRIORITY") ? atoi(getenv("WT_SIDE_PRIORITY")) : 0;   // experiments: 1 high, -1 low
        WT_HIP(hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, prio_env > 0 ? hi : (prio_env < 0 ? lo : (lo + hi) / 2)));
        WT_HIP(hipEventCreateWithFlags(&c->ev_side_done, hipEventDisableTiming));
    }
    WT_HIP(hipStreamWaitEvent(c->side_stream, after, 0));
    std::swap(c->stream, c->side_stream);
    c->in_side = 1;
    c->side_pending = true;
    return 0;
}

void wt_side_end(wt_ctx *c)
{
    std::swap(c->stream, c->side_stream);
    c->in_side = 0;
}

int wt_scale_events(wt_ctx *c, std::vector<hipEvent_t> &ev, int n)
{
    while ((int)ev.size() < n) {
        hipEvent_t e = nullptr;
        WT_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ev.push_back(e);
    }
    return 0;
}

// =============================================================================================
// context
// =============================================================================================
static const int kPartialBlocks = 2048;     // 8 blocks of wt_reduce_kernel per CU

extern "C" int wt_ctx_create(int device, wt_ctx **out)
{
    if (!out) WT_FAIL("wt_ctx_create: null pointer");
    int n = 0;
    WT_HIP(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) WT_FAIL("wt_ctx_create: device %d out of range (%d devices)", device, n);
    WT_HIP(hipSetDevice(device));
    wt_ctx *c = new wt_ctx();
    c->device = device;
    {
        int cus = 0;      // the chunk searches size their grids to the compute units of THIS device
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) c->num_cus = cus;
        else (void)hipGetLastError();
    }
    WT_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    WT_HIP(hipEventCreate(&c->t0));
    WT_HIP(hipEventCreate(&c->t1));
    WT_HIP(hipMalloc(&c->d_hist, (WT_HIST_BINS + 64) * sizeof(uint32_t)));   // bins, float32 select state (+4), float64 state (+16), 64-bit result (+32)
    WT_HIP(hipMalloc(&c->d_partials, (kPartialBlocks * 4 + 8) * sizeof(double)));
    c->partial_blocks = kPartialBlocks;
    WT_HIP(hipHostMalloc(&c->h_pinned, 65536, hipHostMallocDefault));
    WT_HIP(hipMalloc(&c->d_psf, 4096 * sizeof(float)));
    c->d_psf_cap = 4096;
    *out = c;
    return 0;
}

extern "C" int wt_ctx_destroy(wt_ctx *c)
{
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    if (c->comm_stream) {
        (void)hipStreamDestroy(c->comm_stream);
        (void)hipEventDestroy(c->ev_to_comm);
        (void)hipEventDestroy(c->ev_from_comm);
    }
    if (c->xfer_in) {
        (void)hipStreamDestroy(c->xfer_in);
        (void)hipStreamDestroy(c->xfer_out);
    }
    if (c->side_stream) {
        (void)hipStreamSynchronize(c->side_stream);
        (void)hipStreamDestroy(c->side_stream);
        (void)hipEventDestroy(c->ev_side_done);
    }
    for (auto &p : c->pending) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    (void)hipEventDestroy(c->t0);
    (void)hipEventDestroy(c->t1);
    (void)hipFree(c->d_hist);
    (void)hipFree(c->d_partials);
    (void)hipHostFree(c->h_pinned);
    (void)hipFree(c->d_psf);
    if (c->d_taps) (void)hipFree(c->d_taps);
    if (c->d_cand) (void)hipFree(c->d_cand);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

extern "C" int wt_device_memory(wt_ctx *c, int64_t out[2])
{
    WtGuard guard_(ctx_of(c));
    if (!c || !out) WT_FAIL("wt_device_memory: null pointer");
    WT_HIP(hipSetDevice(c->device));
    size_t fr = 0, tot = 0;
    WT_HIP(hipMemGetInfo(&fr, &tot));
    out[0] = (int64_t)fr;
    out[1] = (int64_t)tot;
    return 0;
}

extern "C" int wt_ctx_sync(wt_ctx *c)
{
    WtGuard guard_(ctx_of(c));
    if (!c) WT_FAIL("wt_ctx_sync: null context");
    WT_TRY(wt_side_join(c));
    WT_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int wt_timer_start(wt_ctx *c)
{
    WtGuard guard_(ctx_of(c));
    if (!c) WT_FAIL("wt_timer_start: null context");
    WT_HIP(hipEventRecord(c->t0, c->stream));
    return 0;
}

extern "C" int wt_timer_stop(wt_ctx *c, float *ms)
{
    WtGuard guard_(ctx_of(c));
    if (!c || !ms) WT_FAIL("wt_timer_stop: null pointer");
    WT_TRY(wt_side_join(c));
    WT_HIP(hipEventRecord(c->t1, c->stream));
    WT_HIP(hipEventSynchronize(c->t1));
    WT_HIP(hipEventElapsedTime(ms, c->t0, c->t1));
    return 0;
}

extern "C" int wt_profile_enable(wt_ctx *c, int on)
{
    WtGuard guard_(ctx_of(c));
    if (!c) WT_FAIL("wt_profile_enable: null context");
    WT_TRY(prof_resolve(c));
    c->profiling = on != 0;
    return 0;
}

extern "C" int wt_profile_reset(wt_ctx *c)
{
    WtGuard guard_(ctx_of(c));
    if (!c) WT_FAIL("wt_profile_reset: null context");
    WT_TRY(prof_resolve(c));
    c->prof.clear();
    c->prof_order.clear();
    return 0;
}

extern "C" int wt_profile_count(wt_ctx *c, int *n)
{
    WtGuard guard_(ctx_of(c));
    if (!c || !n) WT_FAIL("wt_profile_count: null pointer");
    WT_TRY(prof_resolve(c));
    *n = (int)c->prof_order.size();
    return 0;
}

extern "C" int wt_profile_entry(wt_ctx *c, int i, char *name64, int64_t *calls, double *total_ms)
{
    WtGuard guard_(ctx_of(c));
    if (!c || !name64 || !calls || !total_ms) WT_FAIL("wt_profile_entry: null pointer");
    WT_TRY(prof_resolve(c));
    if (i < 0 || i >= (int)c->prof_order.size()) WT_FAIL("wt_profile_entry: index %d out of range", i);
    const std::string &nm = c->prof_order[i];
    snprintf(name64, 64, "%s", nm.c_str());
    *calls = c->prof[nm].calls;
    *total_ms = c->prof[nm].ms;
    return 0;
}

// =============================================================================================
// plan
// =============================================================================================
static inline int family_taps(int family) { return family == WT_B3SPLINE ? 5 : 3; }
// A/B switch (wt_set_option "tri4"): four-scale passes of the 3-tap family for level >= 8
static int g_opt_tri4 = getenv("WT_NO_TRI4") ? 0 : 1;
// planes over shuffled physical chunks (plan_alloc): chunks are created in groups worth this many
// planes; 0 = plain hipMalloc per plane (contiguous planes: interop through wt_plane_ptr)
static int g_opt_scatter = getenv("WT_SCATTER") ? atoi(getenv("WT_SCATTER")) : 4;
// strip plans (nranks > 1) too: wt_set_option("scatter_strips", 1) / WT_SCATTER_STRIPS=1; bench.py --gpus N
// measures both placements on the real transport and keeps the faster
static int g_opt_scatter_strips = getenv("WT_SCATTER_STRIPS") ? atoi(getenv("WT_SCATTER_STRIPS")) : 0;

extern "C" int wt_schedule(int family, int level, int fused, int32_t *triples, int cap, int *n_passes)
{
    if (!triples || !n_passes) WT_FAIL("wt_schedule: null pointer");
    if (family != WT_TRIANGLE && family != WT_B3SPLINE) WT_FAIL("wt_schedule: unknown family %d", family);
    if (level < 0 || level > 30) WT_FAIL("wt_schedule: level %d out of range", level);
    const int hw = family_taps(family) / 2;
    int n = 0, s = 0;
    // 3-tap family: passes of FOUR scales (wt_fused.h) - (0,4) and (4,4) from 8 scales on, (0,4)
    // alone for exactly 4 scales (one pass instead of two; at 5 to 7 scales the three-scale passes
    // stay: every pass of the schedule is then a fused one, which is what lets wt_decompose_sum
    // and the interleaved denoise carry the sum); else passes of up to three scales from scales 0
    // and 3 and of two from scale 6
    const bool four = fused && family == WT_TRIANGLE && (level >= 8 || level == 4) && g_opt_tri4;
    while (s < level) {
        int ns = 1;
        if (four && (s == 0 || (s == 4 && level >= 8))) ns = 4;
        else if (four) ns = 1;
        else if (fused && s <= 3) ns = std::min(3, level - s);
        else if (fused && s == 6) ns = std::min(2, level - s);   // D = 64: two scales (x halo hw*3*64)
        if (n >= cap) WT_FAIL("wt_schedule: capacity %d too small", cap);
        triples[3 * n + 0] = s;
        triples[3 * n + 1] = ns;
        triples[3 * n + 2] = hw * ((1 << (s + ns)) - (1 << s));
        ++n;
        s += ns;
    }
    *n_passes = n;
    return 0;
}

// One plane of `need` bytes as a contiguous virtual range over shuffled physical chunks (see
// plan_alloc).  Returns non-zero without leaving a mapping or a fresh handle behind if any step
// fails; `why` then names the failing call.
static int vmm_plane_alloc(wt_plan *p, size_t need, int scatter, void **out, std::string &why, hipError_t &err)
{
    auto fail = [&](const char *call, hipError_t e) {
        err = e;
        char buf[160];
        snprintf(buf, sizeof buf, "%s: HIP error %d (%s)", call, (int)e, hipGetErrorString(e));
        why = buf;
        (void)hipGetLastError();
        return 1;
    };
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = p->ctx->device;
    hipError_t e;
    if (!p->vmm_gran) {
        size_t g = 0;
        e = hipMemGetAllocationGranularity(&g, &prop, hipMemAllocationGranularityRecommended);
        if (e != hipSuccess || g == 0) return fail("hipMemGetAllocationGranularity", e);
        // chunk size: 2 MiB (WT_SCATTER_CHUNK_KB for experiments: chunks below 2 MiB cost TLB reach -
        // 512 KiB: +30 %, 128 KiB: +85 % step time; 8-64 MiB: no different from 2 MiB)
        // Round 5: 8 MiB by default - a quarter of the hipMemCreate / hipMemMap calls of a plan's first use
        // (plan creation 14.8 -> 9 ms at 8192^2), the same step time (tools/first_call.py)
        static const size_t chunk_kb = getenv("WT_SCATTER_CHUNK_KB") ? (size_t)atoll(getenv("WT_SCATTER_CHUNK_KB")) : 8192;
        size_t gmin = 0;
        if (hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gmin == 0) gmin = g;
        p->vmm_gran = std::max<size_t>(gmin, (chunk_kb << 10) / gmin * gmin);
    }
    const size_t g = p->vmm_gran;
    const size_t nchunks = (need + g - 1) / g;
    const size_t size = nchunks * g;
    if (p->vmm_pool.size() < nchunks) {        // refill: chunks for `scatter` planes, shuffled
        const size_t add = nchunks * (size_t)scatter - p->vmm_pool.size();
        std::vector<hipMemGenericAllocationHandle_t> fresh;
        fresh.reserve(add);
        e = hipSuccess;
        for (size_t i = 0; i < add; ++i) {
            hipMemGenericAllocationHandle_t h;
            e = hipMemCreate(&h, g, &prop, 0);
            // out of memory part-way: what we got is enough if it covers THIS plane
            if (e != hipSuccess) break;
            fresh.push_back(h);
        }
        if (p->vmm_pool.size() + fresh.size() < nchunks) {
            // give the fresh chunks back before the caller falls back to hipMalloc for this plane:
            // the fallback must not fail for want of the memory a failed refill is sitting on
            for (auto h : fresh) (void)hipMemRelease(h);
            return fail("hipMemCreate", e);
        }
        uint64_t st = p->vmm_seed;
        for (size_t i = fresh.size(); i > 1; --i) {                     // Fisher-Yates, xorshift stream
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            std::swap(fresh[i - 1], fresh[st % i]);
        }
        p->vmm_seed = st;
        p->vmm_pool.insert(p->vmm_pool.end(), fresh.begin(), fresh.end());
    }
    void *va = nullptr;
    e = hipMemAddressReserve(&va, size, 0, nullptr, 0);
    if (e != hipSuccess) return fail("hipMemAddressReserve", e);
    wt_plan::VmmPlane vp{va, size, {}};
    vp.chunks.reserve(nchunks);
    size_t mapped = 0;
    for (; mapped < nchunks; ++mapped) {
        e = hipMemMap((char *)va + mapped * g, g, 0, p->vmm_pool[p->vmm_pool.size() - 1 - mapped], 0);
        if (e != hipSuccess) break;
    }
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    hipError_t e2 = mapped < nchunks ? e : hipMemSetAccess(va, size, &acc, 1);
    if (e2 != hipSuccess) {
        // unmap chunk by chunk, at the granularity of the hipMemMap calls (one range per handle)
        for (size_t i = 0; i < mapped; ++i) (void)hipMemUnmap((char *)va + i * g, g);
        (void)hipMemAddressFree(va, size);
        return fail(mapped < nchunks ? "hipMemMap" : "hipMemSetAccess", e2);
    }
    for (size_t i = 0; i < nchunks; ++i) vp.chunks.push_back(p->vmm_pool[p->vmm_pool.size() - 1 - i]);
    p->vmm_pool.resize(p->vmm_pool.size() - nchunks);
    p->vmm_planes.push_back(std::move(vp));
    *out = va;
    return 0;
}

// Tear down everything vmm_plane_alloc built.  Every mapping is undone with the granularity it was
// made with (HIP documents hipMemUnmap for whole mappings, not for a range that spans several), and
// every return code is looked at: a rejected unmap / release would leave the physical chunks
// referenced, i.e. leak HBM silently.  Returns the number of failed calls (first one in `why`).
static int vmm_release_all(wt_plan *p, std::string &why)
{
    int bad = 0;
    auto chk = [&](const char *call, hipError_t e) {
        if (e == hipSuccess) return;
        if (!bad++) {
            char buf[160];
            snprintf(buf, sizeof buf, "%s: HIP error %d (%s)", call, (int)e, hipGetErrorString(e));
            why = buf;
        }
        (void)hipGetLastError();
    };
    const size_t g = p->vmm_gran;
    for (auto &v : p->vmm_planes) {
        for (size_t i = 0; i < v.chunks.size(); ++i) {
            chk("hipMemUnmap", hipMemUnmap((char *)v.va + i * g, g));
            chk("hipMemRelease", hipMemRelease(v.chunks[i]));
        }
        chk("hipMemAddressFree", hipMemAddressFree(v.va, v.size));
    }
    p->vmm_planes.clear();
    for (auto h : p->vmm_pool) chk("hipMemRelease", hipMemRelease(h));
    p->vmm_pool.clear();
    return bad;
}

static int plan_alloc(wt_plan *p, float **slot)
{
    if (*slot) return 0;
    WT_HIP(hipSetDevice(p->ctx->device));
    const size_t skew_max = p->skew_floats * 16;
    const size_t need = (p->plane_floats + skew_max) * sizeof(float);
    // WT_ARENA=n (experiment): the first n planes of a plan are carved from ONE allocation, so that
    // their relative placement (and with it the HBM channel / bank relation between the planes a
    // pass writes side by side) does not depend on what the allocator hands out per call
    static const int arena_planes = getenv("WT_ARENA") ? atoi(getenv("WT_ARENA")) : 0;
    void *raw = nullptr;
    if (arena_planes > 0) {
        static const size_t arena_pad = getenv("WT_ARENA_PAD") ? (size_t)atoll(getenv("WT_ARENA_PAD")) / 16 * 16 : 0;
        const size_t stride = (need + 4095) / 4096 * 4096 + arena_pad;
        if (!p->arena) {
            WT_HIP(hipMalloc(&p->arena, stride * (size_t)arena_planes));
            p->raw_allocs.push_back(p->arena);
            p->raw_bytes += stride * (size_t)arena_planes;
            p->arena_left = arena_planes;
            p->arena_stride = stride;
        }
        if (p->arena_left > 0) {
            raw = (char *)p->arena + (size_t)(arena_planes - p->arena_left) * p->arena_stride;
            p->arena_left--;
        }
    }
    // Planes whose physical memory is NOT one contiguous run (default; WT_SCATTER=0 restores plain
    // hipMalloc, WT_SCATTER=c sets the group size).  Measured on MI355X (profiles/r02_d): the same
    // binary runs the headline step in 0.61-0.66 ms when the planes' 2-MiB pages are scattered and in
    // 0.76 ms when the planes lie physically back to back (one arena - or a freshly booted box, whose
    // allocator hands out consecutive blocks: the "slow hosts" of round 1).  The passes write the
    // same pixel of 5 planes side by side; with planes a power of two apart those addresses differ
    // only in bits the HBM channel / bank hash folds away, and the streams fight over the same banks.
    // So: physical chunks of the allocation granularity (2 MiB) are created in groups worth
    // `scatter` planes and dealt to the planes in a shuffled order (fixed seed); each plane stays one
    // contiguous VIRTUAL range (hipMemAddressReserve / hipMemMap).  Small planes (< 8 MiB) stay on
    // hipMalloc: nothing to gain, and a map call per chunk to lose.
    const int scatter = g_opt_scatter;   // wt_set_option("scatter", n); WT_SCATTER sets the initial value
    // Strip plans keep plain hipMalloc unless WT_SCATTER_STRIPS=1: RCCL reads and writes the planes
    // of a strip, and its xGMI transport has never run on mapped memory here (the socket transport
    // of the one-GPU rank test has, green) - the one multi-GPU measurement must not hinge on it.
    if (!raw && scatter > 0 && !p->ctx->vmm_disabled && need >= ((size_t)8 << 20) && (p->nranks == 1 || g_opt_scatter_strips)) {
        std::string why;
        hipError_t err = hipSuccess;
        if (vmm_plane_alloc(p, need, scatter, &raw, why, err)) {
            // Out of memory is transient (this plane takes the hipMalloc path, which reports it if
            // it persists).  Anything else, e.g. hipErrorNotSupported: plain hipMalloc on THIS
            // context from now on.  Not silent: wt_plan_memory reports the state and keeps the
            // reason (WT_VERBOSE prints it) - the headline step is ~20 % slower with the planes
            // physically back to back.
            if (err != hipErrorOutOfMemory) {
                p->ctx->vmm_disabled = true;
                p->ctx->vmm_reason = why;
            }
            // the permanent fallback is announced once per context, unconditionally (the fused passes
            // are ~20 % slower on contiguous planes); transient out-of-memory only under WT_VERBOSE
            if (err != hipErrorOutOfMemory || getenv("WT_VERBOSE"))
                fprintf(stderr, "watroo_hip: plane not scattered on device %d (%s)%s\n", p->ctx->device, why.c_str(),
                        err != hipErrorOutOfMemory ? "; scattered planes disabled for this context: plain hipMalloc from now on" : "");
            raw = nullptr;
        }
    }
    if (!raw) {
        WT_HIP(hipMalloc(&raw, need));
        p->raw_allocs.push_back(raw);
        p->raw_bytes += need;
    }
    *slot = (float *)raw + p->skew_floats * (size_t)(p->n_allocs % 16);
    p->n_allocs++;
    return 0;
}

// pointer to LOCAL ROW 0 of a plane (allocating scratch/out planes on first use)
static int plane_base(wt_plan *p, int id, float **base)
{
    float **slot = nullptr;
    if (id >= 0 && id <= p->max_level) slot = &p->coef[id];
    else if (id == WT_PLANE_INPUT) slot = &p->input;
    else if (id == WT_PLANE_OUT) slot = &p->out;
    else if (id <= WT_PLANE_SCRATCH(0) && id > WT_PLANE_SCRATCH(WT_NUM_SCRATCH)) slot = &p->scratch[-3 - id];
    else WT_FAIL("invalid plane id %d (max_level %d)", id, p->max_level);
    if (p->ctx->prehist_plan == p && p->ctx->prehist_plane == id) p->ctx->prehist_plan = nullptr;   // plane touched
    if (!p->ctx->in_side) {            // a main-stream access: behind everything the side stream has queued
        WT_TRY(wt_side_join(p->ctx));
        p->overlap_ok = false;
    }
    WT_TRY(plan_alloc(p, slot));
    *base = *slot + (size_t)p->g.halo * p->g.P;
    return 0;
}

extern "C" int wt_plan_create_strip(wt_ctx *ctx, int64_t H, int64_t W, int family, int max_level,
                                    int64_t row0, int64_t nrows, int64_t halo_rows, int rank,
                                    int nranks, wt_plan **out)
{
    WtGuard guard_(ctx_of(ctx));
    if (!ctx || !out) WT_FAIL("wt_plan_create: null pointer");
    if (family != WT_TRIANGLE && family != WT_B3SPLINE) WT_FAIL("wt_plan_create: unknown family %d", family);
    if (H < 1 || W < 1 || H > (1 << 30) || W > (1 << 30)) WT_FAIL("wt_plan_create: bad image size %lld x %lld", (long long)H, (long long)W);
    if (max_level < 0 || max_level > 30) WT_FAIL("wt_plan_create: max_level %d out of range", max_level);
    if (row0 < 0 || nrows < 1 || row0 + nrows > H) WT_FAIL("wt_plan_create: strip [%lld,+%lld) outside image of %lld rows", (long long)row0, (long long)nrows, (long long)H);
    if (nranks < 1 || rank < 0 || rank >= nranks) WT_FAIL("wt_plan_create: bad rank %d/%d", rank, nranks);
    if (nranks == 1 && (row0 != 0 || nrows != H)) WT_FAIL("wt_plan_create: a single strip must cover the whole image");
    const int hw = family_taps(family) / 2;
    int64_t halo = 0;
    if (nranks > 1) {
        halo = halo_rows > 0 ? halo_rows : (max_level > 0 ? (int64_t)hw << (max_level - 1) : 0);
        // the fused schedule needs the cumulative halo of its widest pass
        int32_t tr[3 * 32];
        int np = 0;
        WT_TRY(wt_schedule(family, max_level, 1, tr, 32, &np));
        for (int i = 0; i < np; ++i) halo = std::max<int64_t>(halo, tr[3 * i + 2]);
    }
    // WT_PITCH_PAD (pixels, multiple of 4): extra row pitch for experiments with the HBM channel
    // mapping of row-marching kernels
    static const int64_t pitch_pad = getenv("WT_PITCH_PAD") ? std::max<int64_t>(0, atoll(getenv("WT_PITCH_PAD")) / 4 * 4) : 0;
    const int64_t P = (W + 3) / 4 * 4 + pitch_pad;
    // Kernels index rows / columns with int32 and pixels with 64-bit offsets, but the flat pointwise
    // kernels count float4 groups in int64 and the tests cover planes up to 2^30 pixels (32768^2):
    // larger strips are refused rather than run unverified.
    if ((nrows + 2 * halo) * P > ((int64_t)1 << 31))
        WT_FAIL("wt_plan_create: a strip of %lld x %lld pixels (incl. margins) exceeds 2^31 per plane; split it into more strips",
                (long long)(nrows + 2 * halo), (long long)P);
    wt_plan *p = new wt_plan();
    p->ctx = ctx;
    p->g = Geo{(int)W, (int)P, (int)H, (int)row0, (int)nrows, (int)halo, 0};
    p->family = family;
    p->max_level = max_level;
    p->rank = rank;
    p->nranks = nranks;
    p->plane_floats = (size_t)(nrows + 2 * halo) * (size_t)P;
    {
        // default skew: 4 KiB + 256 B per plane index (keeps 16-byte alignment); WT_PLANE_SKEW
        // (bytes, multiple of 16) overrides it for experiments
        const char *e = getenv("WT_PLANE_SKEW");
        size_t skew_bytes = e ? (size_t)atoll(e) : 4352;
        p->skew_floats = (skew_bytes / 16 * 16) / 4;
    }
    p->coef.assign(max_level + 1, nullptr);
    for (int i = 0; i <= max_level; ++i) {
        int rc = plan_alloc(p, &p->coef[i]);
        if (rc) {
            wt_plan_destroy(p);
            return rc;
        }
    }
    *out = p;
    return 0;
}

extern "C" int wt_plan_create(wt_ctx *ctx, int64_t H, int64_t W, int family, int max_level, wt_plan **out)
{
    WtGuard guard_(ctx_of(ctx));
    return wt_plan_create_strip(ctx, H, W, family, max_level, 0, H, 0, 0, 1, out);
}

static void destroy_events(std::vector<hipEvent_t> &ev)
{
    for (auto e : ev) (void)hipEventDestroy(e);
    ev.clear();
}

extern "C" int wt_plan_destroy(wt_plan *p)
{
    WtGuard guard_(ctx_of(p));
    if (!p) return 0;
    if (p->ctx->prehist_plan == p) p->ctx->prehist_plan = nullptr;
    (void)hipSetDevice(p->ctx->device);
    (void)wt_side_join(p->ctx);
    (void)hipStreamSynchronize(p->ctx->stream);
    destroy_events(p->scale_ev);
    int bad = 0;
    std::string why;
    for (void *q : p->raw_allocs) {
        hipError_t e = hipFree(q);
        if (e != hipSuccess && !bad++) why = std::string("hipFree: ") + hipGetErrorString(e);
    }
    bad += vmm_release_all(p, why);
    delete p;
    if (bad) WT_FAIL("wt_plan_destroy: %d release call(s) failed, device memory may still be held (%s)", bad, why.c_str());
    return 0;
}

extern "C" int wt_plan_memory(wt_plan *p, int64_t out[4])
{
    WtGuard guard_(ctx_of(p));
    if (!p || !out) WT_FAIL("wt_plan_memory: null pointer");
    size_t mapped = 0;
    for (auto &v : p->vmm_planes) mapped += v.size;
    const size_t idle = p->vmm_pool.size() * p->vmm_gran;
    out[0] = (int64_t)(p->raw_bytes + mapped + idle);
    out[1] = (int64_t)mapped;
    out[2] = (int64_t)idle;
    out[3] = p->ctx->vmm_disabled ? 1 : 0;      // why: wt_ctx_scatter_status
    return 0;
}

// Whether the planes of this context's plans are still mapped over scattered chunks, and if not, the
// call that made the context fall back to plain hipMalloc (kept from the moment it happened; the
// thread's wt_last_error is left alone).
extern "C" int wt_ctx_scatter_status(wt_ctx *c, int *disabled, char *reason, int cap)
{
    WtGuard guard_(ctx_of(c));
    if (!c || !disabled) WT_FAIL("wt_ctx_scatter_status: null pointer");
    *disabled = c->vmm_disabled ? 1 : 0;
    if (reason && cap > 0) snprintf(reason, (size_t)cap, "%s", c->vmm_reason.c_str());
    return 0;
}

extern "C" int wt_plan_trim(wt_plan *p)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_plan_trim: null plan");
    int bad = 0;
    hipError_t first = hipSuccess;
    for (auto h : p->vmm_pool) {
        hipError_t e = hipMemRelease(h);
        if (e != hipSuccess && !bad++) first = e;
    }
    p->vmm_pool.clear();
    if (bad) WT_FAIL("wt_plan_trim: hipMemRelease failed %d time(s) (%s)", bad, hipGetErrorString(first));
    return 0;
}

extern "C" int wt_plan_info(wt_plan *p, int64_t out[8])
{
    WtGuard guard_(ctx_of(p));
    if (!p || !out) WT_FAIL("wt_plan_info: null pointer");
    out[0] = p->g.H; out[1] = p->g.W; out[2] = p->g.P; out[3] = p->g.row0;
    out[4] = p->g.nrows; out[5] = p->g.halo; out[6] = p->max_level; out[7] = p->family;
    return 0;
}

extern "C" int wt_plan_set_border(wt_plan *p, int border)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_plan_set_border: null plan");
    if (border < 0 || border > 3) WT_FAIL("wt_plan_set_border: unknown border mode %d", border);
    if (border && p->nranks > 1) WT_FAIL("wt_plan_set_border: non-default borders are single-GPU only");
    p->g.border = border;
    return 0;
}

// Planes built from scattered physical chunks (plan_alloc): the memcpy engines refuse ranges that
// span several mapped handles, so host transfers bounce through a hipMalloc'ed plane and every
// device-to-device copy of plane data is a kernel.
static bool is_vmm(const wt_plan *p, const float *b)
{
    for (auto &v : p->vmm_planes)
        if ((const char *)b >= (const char *)v.va && (const char *)b < (const char *)v.va + v.size) return true;
    return false;
}
static int vmm_stage(wt_plan *p, float **stage)
{
    if (!p->vmm_stage) {
        void *raw = nullptr;
        WT_HIP(hipMalloc(&raw, p->plane_floats * sizeof(float)));
        p->raw_allocs.push_back(raw);
        p->raw_bytes += p->plane_floats * sizeof(float);
        p->vmm_stage = (float *)raw;
    }
    *stage = p->vmm_stage + (size_t)p->g.halo * p->g.P;
    return 0;
}
static int vmm_copy(wt_plan *p, float *dst, const float *src)
{
    const int64_t n4 = (int64_t)p->g.nrows * p->g.P / 4;
    hipLaunchKernelGGL(wt_copy_kernel, dim3((unsigned)std::min<int64_t>((n4 + 255) / 256, 2048)), dim3(256), 0, p->ctx->stream, dst, src, n4);
    WT_HIP(hipGetLastError());
    return 0;
}
// rows x cols floats, device to device, on `st`
static int copy2d(wt_plan *a, wt_plan *b, float *dst, size_t dpitch, const float *src, size_t spitch, size_t cols,
                  size_t rows, hipStream_t st)
{
    if (rows == 0 || cols == 0) return 0;
    if (is_vmm(a, dst) || is_vmm(a, src) || is_vmm(b, dst) || is_vmm(b, src)) {
        dim3 grid((unsigned)std::min<size_t>((cols + 255) / 256, 64), (unsigned)std::min<size_t>(rows, 4096));
        hipLaunchKernelGGL(wt_copy2d_kernel, grid, dim3(256), 0, st, dst, (int64_t)dpitch, src, (int64_t)spitch, (int)cols, (int)rows);
        WT_HIP(hipGetLastError());
        return 0;
    }
    WT_HIP(hipMemcpy2DAsync(dst, dpitch * 4, src, spitch * 4, cols * 4, rows, hipMemcpyDeviceToDevice, st));
    return 0;
}

// dst_plane of `dst` <- the dst-sized window of src_plane of `src` starting at (y0, x0)
extern "C" int wt_plan_set_taps(wt_plan *p, const float *taps, int ntaps)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_plan_set_taps: null plan");
    if (ntaps == 0) { p->ntaps = 0; return 0; }
    if (!taps) WT_FAIL("wt_plan_set_taps: null taps");
    if (ntaps < 1 || ntaps > WT_MAX_CUSTOM_TAPS || !(ntaps & 1))
        WT_FAIL("wt_plan_set_taps: %d taps unsupported (odd, 1..%d)", ntaps, WT_MAX_CUSTOM_TAPS);
    if (p->nranks > 1) WT_FAIL("wt_plan_set_taps: user-defined scaling functions are single-GPU only");
    for (int i = 0; i < ntaps; ++i) p->taps[i] = taps[i];
    p->ntaps = ntaps;
    return 0;
}

static CustomTaps plan_taps(const wt_plan *p)
{
    CustomTaps t{};
    t.n = p->ntaps;
    for (int i = 0; i < p->ntaps; ++i) t.k[i] = p->taps[i];
    return t;
}

// separable filter with the plan's run-time taps: rows into scratch 15, then columns (+ detail)
static int launch_custom(wt_plan *p, const float *in, float *out_c, float *out_w, int s, int square, const char *name)
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

static inline int64_t plan_n4(const wt_plan *p);
static inline int flat_grid(int64_t n4);
// sdev_loc (watroo/wavelets.py:24-32) with run-time taps: both moments through the generic
// separable kernels (mean in scratch 13), then the clip / sqrt / factors
static int launch_custom_variance(wt_plan *p, const float *in, float *out, int s, float f1, float f2, int take_sqrt,
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

extern "C" int wt_crop_plane(wt_plan *src, int src_plane, wt_plan *dst, int dst_plane, int64_t y0, int64_t x0)
{
    WtGuard guard_(ctx_of(src), ctx_of(dst));
    if (!src || !dst) WT_FAIL("wt_crop_plane: null plan");
    if (src->ctx->device != dst->ctx->device) WT_FAIL("wt_crop_plane: plans on different devices");
    if (y0 < 0 || x0 < 0 || y0 + dst->g.nrows > src->g.nrows || x0 + dst->g.W > src->g.W)
        WT_FAIL("wt_crop_plane: window outside the source plane");
    float *s_ = nullptr, *d_ = nullptr;
    WT_TRY(plane_base(src, src_plane, &s_));
    WT_TRY(plane_base(dst, dst_plane, &d_));
    WT_TRY(copy2d(src, dst, d_, (size_t)dst->g.P, s_ + (size_t)y0 * src->g.P + x0, (size_t)src->g.P, (size_t)dst->g.W,
                  (size_t)dst->g.nrows, src->ctx->stream));
    if (dst->ctx->stream != src->ctx->stream) WT_HIP(hipStreamSynchronize(src->ctx->stream));
    return 0;
}

extern "C" int wt_paste_plane(wt_plan *src, int src_plane, wt_plan *dst, int dst_plane, int64_t y0, int64_t x0)
{
    WtGuard guard_(ctx_of(src), ctx_of(dst));
    if (!src || !dst) WT_FAIL("wt_paste_plane: null plan");
    if (src->ctx->device != dst->ctx->device) WT_FAIL("wt_paste_plane: plans on different devices");
    if (y0 < 0 || x0 < 0 || y0 + src->g.nrows > dst->g.nrows || x0 + src->g.W > dst->g.W)
        WT_FAIL("wt_paste_plane: window outside the destination plane");
    float *s_ = nullptr, *d_ = nullptr;
    WT_TRY(plane_base(src, src_plane, &s_));
    WT_TRY(plane_base(dst, dst_plane, &d_));
    WT_TRY(copy2d(src, dst, d_ + (size_t)y0 * dst->g.P + x0, (size_t)dst->g.P, s_, (size_t)src->g.P, (size_t)src->g.W,
                  (size_t)src->g.nrows, src->ctx->stream));
    if (dst->ctx->stream != src->ctx->stream) WT_HIP(hipStreamSynchronize(src->ctx->stream));
    return 0;
}

// dst[dy:dy+rows, dx:dx+cols] = src[sy:sy+rows, sx:sx+cols]  (local rows; same device)
extern "C" int wt_copy_window(wt_plan *src, int src_plane, wt_plan *dst, int dst_plane, int64_t sy, int64_t sx,
                              int64_t dy, int64_t dx, int64_t rows, int64_t cols)
{
    WtGuard guard_(ctx_of(src), ctx_of(dst));
    if (!src || !dst) WT_FAIL("wt_copy_window: null plan");
    if (src->ctx->device != dst->ctx->device) WT_FAIL("wt_copy_window: plans on different devices");
    if (rows < 1 || cols < 1 || sy < 0 || sx < 0 || dy < 0 || dx < 0 || sy + rows > src->g.nrows || sx + cols > src->g.W ||
        dy + rows > dst->g.nrows || dx + cols > dst->g.W)
        WT_FAIL("wt_copy_window: window outside a plane");
    float *s_ = nullptr, *d_ = nullptr;
    WT_TRY(plane_base(src, src_plane, &s_));
    WT_TRY(plane_base(dst, dst_plane, &d_));
    if (s_ == d_) WT_FAIL("wt_copy_window: source and destination are the same plane");
    WT_TRY(copy2d(src, dst, d_ + (size_t)dy * dst->g.P + dx, (size_t)dst->g.P, s_ + (size_t)sy * src->g.P + sx,
                  (size_t)src->g.P, (size_t)cols, (size_t)rows, src->ctx->stream));
    if (dst->ctx->stream != src->ctx->stream) WT_HIP(hipStreamSynchronize(src->ctx->stream));
    return 0;
}

extern "C" int wt_plane_ptr(wt_plan *p, int plane, void **dev_ptr)
{
    WtGuard guard_(ctx_of(p));
    if (!p || !dev_ptr) WT_FAIL("wt_plane_ptr: null pointer");
    float *b = nullptr;
    WT_TRY(plane_base(p, plane, &b));
    *dev_ptr = b;
    return 0;
}

// =============================================================================================
// host <-> device, copies
// =============================================================================================
// Host transfers.  Caller-owned (pageable) buffers are handed to the runtime as they are: it locks
// the pages for the duration of a copy by itself and reaches the same 56-57 GB/s as page-locked
// memory on this platform (tools/bench_pcie_pipe.py).  Rounds 1-2 additionally registered large
// user buffers for the duration of the call (hipHostRegister / hipHostUnregister around the
// copy); that is OFF by default since round 3: with it a long randomised run (tools/fuzz.py, 140
// cases, multi-megabyte numpy arrays carved from the C heap once glibc has raised its mmap
// threshold) ended twice in "Memory access fault by GPU ... on address <host heap address>" at
// different places, and ran clean twice without it - registering and unregistering ranges of the
// process heap that the allocator later trims or hands out again is not something the runtime
// tolerates.  WT_PIN_THRESHOLD=<bytes> switches the registration back on for experiments.
static size_t pin_threshold()
{
    static const long long v = getenv("WT_PIN_THRESHOLD") ? atoll(getenv("WT_PIN_THRESHOLD")) : 0;
    return (size_t)v;
}

static bool try_pin(const void *host, size_t bytes)
{
    const size_t thr = pin_threshold();
    if (thr == 0 || bytes < thr) return false;
    if (hipHostRegister(const_cast<void *>(host), bytes, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();   // already registered / not registrable: fall back to pageable
        return false;
    }
    return true;
}

// Page-locked host blocks (the result arrays of the numpy-to-numpy calls).  hipHostMalloc of 256 MiB takes 44 ms
// on the MI355X boxes - by far the largest part of the ~100 ms a process's FIRST denoise(img) at 8192^2 took
// (tools/first_call.py) - because it faults and locks 65 536 small pages one by one.  Large blocks are instead
// (round 5) an anonymous mapping with the transparent-huge-page hint, first touched by several threads at once
// (2.5 ms for 256 MiB: 128 huge pages) and then registered with the runtime (hipHostRegister of a faulted
// huge-page range: 0.5 ms); the mapping is ours alone, unregistered before it is unmapped (nothing like the heap
// ranges of the note above).  Small blocks, or any step of this failing: hipHostMalloc as before.
struct WtHostBlock {
    void *map;
    size_t map_bytes;
};
static std::mutex g_host_mu;
static std::map<void *, WtHostBlock> g_host_blocks;     // registered mappings by the pointer handed out
static const size_t kHugePage = (size_t)2 << 20;

static void *host_block_mmap(size_t bytes, WtHostBlock &blk)
{
    const size_t sz = (bytes + kHugePage - 1) / kHugePage * kHugePage;
    void *m = mmap(nullptr, sz + kHugePage, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) return nullptr;
    char *al = (char *)(((uintptr_t)m + kHugePage - 1) & ~(uintptr_t)(kHugePage - 1));
    (void)madvise(al, sz, MADV_HUGEPAGE);
    // first touch in parallel: one write per small page (one fault per huge page where the hint is honoured)
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t nthreads = std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)hw, sz / ((size_t)8 << 20)}));
    const size_t slice = (sz / nthreads + kHugePage - 1) / kHugePage * kHugePage;
    std::vector<std::thread> th;
    for (size_t t = 0; t < nthreads; ++t) {
        const size_t lo = t * slice, hi = std::min(sz, lo + slice);
        if (lo >= hi) break;
        th.emplace_back([al, lo, hi] {
            for (size_t o = lo; o < hi; o += 4096) ((volatile char *)al)[o] = 0;
        });
    }
    for (auto &t : th) t.join();
    if (hipHostRegister(al, sz, hipHostRegisterPortable) != hipSuccess) {
        (void)hipGetLastError();
        (void)munmap(m, sz + kHugePage);
        return nullptr;
    }
    blk = WtHostBlock{m, sz + kHugePage};
    return al;
}

extern "C" int wt_host_alloc(wt_ctx *c, size_t bytes, void **host_ptr)
{
    WtGuard guard_(ctx_of(c));
    if (!c || !host_ptr) WT_FAIL("wt_host_alloc: null pointer");
    if (bytes == 0) WT_FAIL("wt_host_alloc: zero bytes");
    *host_ptr = nullptr;
    WT_HIP(hipSetDevice(c->device));
    static const bool thp_blocks = !getenv("WT_NO_THP_HOST_BLOCKS");
    if (thp_blocks && bytes >= ((size_t)16 << 20)) {
        WtHostBlock blk{};
        if (void *q = host_block_mmap(bytes, blk)) {
            std::lock_guard<std::mutex> lk(g_host_mu);
            g_host_blocks[q] = blk;
            *host_ptr = q;
            return 0;
        }
    }
    WT_HIP(hipHostMalloc(host_ptr, bytes, hipHostMallocPortable));
    return 0;
}

extern "C" int wt_host_free(void *host_ptr)
{
    if (!host_ptr) return 0;
    WtHostBlock blk{};
    bool ours = false;
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        auto it = g_host_blocks.find(host_ptr);
        if (it != g_host_blocks.end()) {
            blk = it->second;
            g_host_blocks.erase(it);
            ours = true;
        }
    }
    if (ours) {
        const hipError_t e = hipHostUnregister(host_ptr);
        (void)munmap(blk.map, blk.map_bytes);
        WT_HIP(e);
        return 0;
    }
    WT_HIP(hipHostFree(host_ptr));
    return 0;
}

extern "C" int wt_upload(wt_plan *p, int plane, const float *host, int64_t host_stride)
{
    WtGuard guard_(ctx_of(p));
    if (!p || !host) WT_FAIL("wt_upload: null pointer");
    if (host_stride < p->g.W) WT_FAIL("wt_upload: host stride %lld < width %d", (long long)host_stride, p->g.W);
    float *b = nullptr;
    WT_TRY(plane_base(p, plane, &b));
    const size_t span = ((size_t)(p->g.nrows - 1) * (size_t)host_stride + (size_t)p->g.W) * 4;
    float *target = b;
    // (a plane mapped over scattered chunks takes host transfers through the hipMalloc'ed bounce plane: the 2-D
    //  copy does not cross mapped chunks, and - tried in round 5 - neither does the flat hipMemcpyAsync of rows
    //  that are contiguous on both sides: half the rate, and a download that silently delivered zeros)
    if (is_vmm(p, b)) WT_TRY(vmm_stage(p, &target));
    const bool pinned = try_pin(host, span);
    hipError_t e = hipMemcpy2DAsync(target, (size_t)p->g.P * 4, host, (size_t)host_stride * 4, (size_t)p->g.W * 4,
                                    (size_t)p->g.nrows, hipMemcpyHostToDevice, p->ctx->stream);
    if (e == hipSuccess && target != b) e = vmm_copy(p, b, target) ? hipErrorUnknown : hipSuccess;
    if (e == hipSuccess) e = hipStreamSynchronize(p->ctx->stream);
    if (pinned) (void)hipHostUnregister(const_cast<float *>(host));
    WT_HIP(e);
    return 0;
}

// plane <- (float) of an image of another element type (uint8 pictures, raw big-endian FITS integers ...:
// everything the reference does NOT recast to float64 and this engine serves in float32), widened on the
// device instead of by a host astype.  Same type codes as wt64_upload_int.
template <typename I>
static void from_elems_launch(wt_plan *p, float *b, bool swap)
{
    const dim3 grid((p->g.W + 255) / 256, (unsigned)std::min(p->g.nrows, 32768)), block(256);
    if (swap) hipLaunchKernelGGL((wt_from_elems_kernel<I, float, true>), grid, block, 0, p->ctx->stream, (const I *)p->istage, b, p->g.W, p->g.P, p->g.nrows);
    else hipLaunchKernelGGL((wt_from_elems_kernel<I, float, false>), grid, block, 0, p->ctx->stream, (const I *)p->istage, b, p->g.W, p->g.P, p->g.nrows);
}

extern "C" int wt_upload_int(wt_plan *p, int plane, const void *host, int64_t host_pitch_bytes, int dtype)
{
    WtGuard guard_(ctx_of(p));
    if (!p || !host) WT_FAIL("wt_upload_int: null pointer");
    static const int isz[11] = {0, 1, 1, 2, 2, 4, 4, 8, 8, 4, 8};
    const bool swap = (dtype & WT_BYTESWAPPED) != 0;
    const int base = dtype & ~WT_BYTESWAPPED;
    if (base < WT_INT8 || base > WT_FLOAT64) WT_FAIL("wt_upload_int: unknown element type %d", dtype);
    const size_t row = (size_t)p->g.W * isz[base];
    if (host_pitch_bytes < (int64_t)row) WT_FAIL("wt_upload_int: row pitch %lld below the %zu bytes of a row", (long long)host_pitch_bytes, row);
    float *b = nullptr;
    WT_TRY(plane_base(p, plane, &b));
    const size_t need = row * p->g.nrows;
    if (p->istage_cap < need) {
        WT_HIP(hipSetDevice(p->ctx->device));
        WT_HIP(hipStreamSynchronize(p->ctx->stream));
        if (p->istage) {
            (void)hipFree(p->istage);
            p->raw_allocs.erase(std::remove(p->raw_allocs.begin(), p->raw_allocs.end(), p->istage), p->raw_allocs.end());
            p->raw_bytes -= p->istage_cap;
            p->istage = nullptr;
            p->istage_cap = 0;
        }
        WT_HIP(hipMalloc(&p->istage, need));
        p->raw_allocs.push_back(p->istage);
        p->raw_bytes += need;
        p->istage_cap = need;
    }
    const bool pinned = try_pin(host, (size_t)(p->g.nrows - 1) * (size_t)host_pitch_bytes + row);
    hipError_t e = hipMemcpy2DAsync(p->istage, row, host, (size_t)host_pitch_bytes, row, p->g.nrows, hipMemcpyHostToDevice, p->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->ctx->stream);          // (the host rows are free again)
    if (pinned) (void)hipHostUnregister(const_cast<void *>(host));
    WT_HIP(e);
    switch (base) {
        case WT_INT8: from_elems_launch<int8_t>(p, b, false); break;
        case WT_UINT8: from_elems_launch<uint8_t>(p, b, false); break;
        case WT_INT16: from_elems_launch<int16_t>(p, b, swap); break;
        case WT_UINT16: from_elems_launch<uint16_t>(p, b, swap); break;
        case WT_INT32: from_elems_launch<int32_t>(p, b, swap); break;
        case WT_UINT32: from_elems_launch<uint32_t>(p, b, swap); break;
        case WT_INT64: from_elems_launch<int64_t>(p, b, swap); break;
        case WT_UINT64: from_elems_launch<uint64_t>(p, b, swap); break;
        case WT_FLOAT32: from_elems_launch<float>(p, b, swap); break;
        default: from_elems_launch<double>(p, b, swap); break;
    }
    WT_HIP(hipGetLastError());
    WT_HIP(hipStreamSynchronize(p->ctx->stream));
    return 0;
}

extern "C" int wt_download(wt_plan *p, int plane, float *host, int64_t host_stride)
{
    WtGuard guard_(ctx_of(p));
    if (!p || !host) WT_FAIL("wt_download: null pointer");
    if (host_stride < p->g.W) WT_FAIL("wt_download: host stride %lld < width %d", (long long)host_stride, p->g.W);
    float *b = nullptr;
    WT_TRY(plane_base(p, plane, &b));
    const size_t span = ((size_t)(p->g.nrows - 1) * (size_t)host_stride + (size_t)p->g.W) * 4;
    const bool pinned = try_pin(host, span);
    if (is_vmm(p, b)) {
        float *stage = nullptr;
        WT_TRY(vmm_stage(p, &stage));
        WT_TRY(vmm_copy(p, stage, b));
        b = stage;
    }
    hipError_t e = hipMemcpy2DAsync(host, (size_t)host_stride * 4, b, (size_t)p->g.P * 4, (size_t)p->g.W * 4,
                                    (size_t)p->g.nrows, hipMemcpyDeviceToHost, p->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(p->ctx->stream);
    if (pinned) (void)hipHostUnregister(host);
    WT_HIP(e);
    return 0;
}

static inline int64_t plan_n4(const wt_plan *p) { return (int64_t)p->g.nrows * p->g.P / 4; }
static inline int flat_grid(int64_t n4) { return (int)std::min<int64_t>((n4 + 255) / 256, 256 * 8); }

extern "C" int wt_copy_plane(wt_plan *p, int src, int dst)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_copy_plane: null plan");
    float *s = nullptr, *d = nullptr;
    WT_TRY(plane_base(p, src, &s));
    WT_TRY(plane_base(p, dst, &d));
    if (s == d) return 0;
    if (is_vmm(p, s) || is_vmm(p, d)) return vmm_copy(p, d, s);
    WT_HIP(hipMemcpyAsync(d, s, (size_t)p->g.nrows * p->g.P * 4, hipMemcpyDeviceToDevice, p->ctx->stream));
    return 0;
}

extern "C" int wt_fill_plane(wt_plan *p, int plane, float value)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_fill_plane: null plan");
    float *d = nullptr;
    WT_TRY(plane_base(p, plane, &d));
    const int64_t n4 = plan_n4(p);
    ProfScope ps(p->ctx, "wt_fill_kernel");
    hipLaunchKernelGGL(wt_fill_kernel, dim3(flat_grid(n4)), dim3(256), 0, p->ctx->stream, d, n4, value);
    WT_HIP(hipGetLastError());
    return 0;
}

// =============================================================================================
// halo exchange
// =============================================================================================
extern "C" int wt_halo_exchange_local(wt_plan *upper, wt_plan *lower, int plane, int64_t rows)
{
    WtGuard guard_(ctx_of(upper), ctx_of(lower));
    if (!upper || !lower) WT_FAIL("wt_halo_exchange_local: null plan");
    if (rows == 0) return 0;
    if (upper->g.P != lower->g.P || upper->g.row0 + upper->g.nrows != lower->g.row0)
        WT_FAIL("wt_halo_exchange_local: plans are not vertically adjacent strips of one image");
    if (rows < 0 || rows > upper->g.halo || rows > lower->g.halo || rows > upper->g.nrows || rows > lower->g.nrows)
        WT_FAIL("wt_halo_exchange_local: %lld rows exceed halo/strip size", (long long)rows);
    float *u = nullptr, *l = nullptr;
    WT_TRY(plane_base(upper, plane, &u));
    WT_TRY(plane_base(lower, plane, &l));
    const size_t P = (size_t)upper->g.P, bytes = (size_t)rows * P * 4;
    hipStream_t st = upper->ctx->stream;
    // upper's last rows -> lower's top margin ; lower's first rows -> upper's bottom margin
    (void)bytes;
    WT_TRY(copy2d(upper, lower, l - (size_t)rows * P, P, u + (size_t)(upper->g.nrows - rows) * P, P, P, (size_t)rows, st));
    WT_TRY(copy2d(upper, lower, u + (size_t)upper->g.nrows * P, P, l, P, P, (size_t)rows, st));
    if (lower->ctx->stream != st) WT_HIP(hipStreamSynchronize(st));
    return 0;
}

// st == nullptr: the context's compute stream
static int halo_exchange_on(wt_plan *p, int plane, int64_t rows, hipStream_t st, const char *prof_name = "rccl_halo_exchange")
{
    if (!p) WT_FAIL("wt_halo_exchange: null plan");
    if (p->nranks == 1 || rows == 0) return 0;
    wt_ctx *c = p->ctx;
    if (!st) st = c->stream;
    if (!c->comm) WT_FAIL("wt_halo_exchange: context has no RCCL communicator (wt_ctx_comm_init)");
    if (c->nranks != p->nranks || c->rank != p->rank) WT_FAIL("wt_halo_exchange: plan rank %d/%d != communicator rank %d/%d", p->rank, p->nranks, c->rank, c->nranks);
    if (rows < 0 || rows > p->g.halo) WT_FAIL("wt_halo_exchange: %lld rows exceed the plan's halo margin %d", (long long)rows, p->g.halo);
    if (rows > p->g.nrows) WT_FAIL("wt_halo_exchange: halo of %lld rows spans more than one neighbour (strip has %d rows)", (long long)rows, p->g.nrows);
    float *b = nullptr;
    WT_TRY(plane_base(p, plane, &b));
    const size_t P = (size_t)p->g.P, cnt = (size_t)rows * P;
    const int up = p->rank - 1, dn = p->rank + 1;
    ProfScope ps(c, prof_name, st);
    WtRcclGroup<RcclApi> grp(g_rccl);      // always closed, also when a Send / Recv fails (wt_rccl_group.h)
    if (up >= 0) {
        grp.run("ncclSend(up)", [&] { return g_rccl.Send(b, cnt, NCCL_FLOAT32, up, c->comm, st); });
        grp.run("ncclRecv(up)", [&] { return g_rccl.Recv(b - cnt, cnt, NCCL_FLOAT32, up, c->comm, st); });
    }
    if (dn < p->nranks) {
        grp.run("ncclSend(down)", [&] { return g_rccl.Send(b + (size_t)(p->g.nrows - rows) * P, cnt, NCCL_FLOAT32, dn, c->comm, st); });
        grp.run("ncclRecv(down)", [&] { return g_rccl.Recv(b + (size_t)p->g.nrows * P, cnt, NCCL_FLOAT32, dn, c->comm, st); });
    }
    if (const int rc = grp.end()) {
        wt_set_error("RCCL error %d (%s) in the halo exchange of plane %d (%lld rows, rank %d/%d): %s", rc,
                     g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?", plane, (long long)rows, p->rank, p->nranks, grp.what);
        return 3;
    }
    return 0;
}

extern "C" int wt_halo_exchange(wt_plan *p, int plane, int64_t rows)
{
    WtGuard guard_(ctx_of(p));
    return halo_exchange_on(p, plane, rows, nullptr);
}

extern "C" int wt_comm_selftest(wt_ctx *c, int64_t nfloats, int *ok)
{
    WtGuard guard_(ctx_of(c));
    if (!c || !ok) WT_FAIL("wt_comm_selftest: null pointer");
    if (!c->comm) WT_FAIL("wt_comm_selftest: no communicator");
    if (nfloats < 1) WT_FAIL("wt_comm_selftest: nfloats must be positive");
    *ok = 0;
    WT_HIP(hipSetDevice(c->device));
    struct DevBuf {          // freed on every return path
        float *p = nullptr;
        ~DevBuf() { if (p) (void)hipFree(p); }
    } abuf, bbuf;
    WT_HIP(hipMalloc(&abuf.p, nfloats * 4));
    WT_HIP(hipMalloc(&bbuf.p, nfloats * 4));
    float *a = abuf.p, *b = bbuf.p;
    std::vector<float> h(nfloats), r(nfloats, 0.f);
    for (int64_t i = 0; i < nfloats; ++i) h[i] = (float)(i % 977) * 0.5f + (float)c->rank;
    WT_HIP(hipMemcpyAsync(a, h.data(), nfloats * 4, hipMemcpyHostToDevice, c->stream));
    WT_HIP(hipMemsetAsync(b, 0, nfloats * 4, c->stream));   // ordered before the Recv into b
    // ring: send to (rank+1)%n, receive from (rank-1+n)%n  (self when n == 1)
    const int to = (c->rank + 1) % c->nranks, from = (c->rank + c->nranks - 1) % c->nranks;
    {
        WtRcclGroup<RcclApi> grp(g_rccl);
        grp.run("ncclSend", [&] { return g_rccl.Send(a, nfloats, NCCL_FLOAT32, to, c->comm, c->stream); });
        grp.run("ncclRecv", [&] { return g_rccl.Recv(b, nfloats, NCCL_FLOAT32, from, c->comm, c->stream); });
        if (const int rc = grp.end()) {
            (void)hipStreamSynchronize(c->stream);      // the buffers are released on return
            wt_set_error("RCCL error %d (%s) in wt_comm_selftest: %s", rc, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?", grp.what);
            return 3;
        }
    }
    // all-reduce of a tiny vector (d_hist is also where a fused first pass leaves the first level of
    // the median select: that marker does not survive this)
    c->prehist_plan = nullptr;
    WT_HIP(hipMemsetAsync(c->d_hist, 0, 16, c->stream));
    WT_NCCL(g_rccl.AllReduce(c->d_hist, c->d_hist, 4, NCCL_UINT32, NCCL_SUM, c->comm, c->stream));
    WT_HIP(hipMemcpyAsync(r.data(), b, nfloats * 4, hipMemcpyDeviceToHost, c->stream));
    WT_HIP(hipStreamSynchronize(c->stream));
    int good = 1;
    for (int64_t i = 0; i < nfloats; ++i)
        if (r[i] != (float)(i % 977) * 0.5f + (float)from) { good = 0; break; }
    *ok = good;
    return 0;
}

// =============================================================================================
// chain-march launches (generic per-scale operator)
// =============================================================================================
static int check_scale(const wt_plan *p, int s, const char *who)
{
    if (s < 0 || s > 24) WT_FAIL("%s: scale %d out of range", who, s);
    const int hw = family_taps(p->family) / 2;
    const int64_t halo = (int64_t)hw << s;
    if (p->nranks > 1) {
        if (halo > p->g.halo) WT_FAIL("%s: scale %d needs %lld halo rows, plan has %d", who, s, (long long)halo, p->g.halo);
    }
    return 0;
}

static inline StencilCtx stencil_ctx(const wt_plan *p, hipStream_t st = nullptr)
{
    return StencilCtx{p->ctx, st ? st : p->ctx->stream, p->g, p->family};
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
static int g_opt_bilateral2 = getenv("WT_NO_BILATERAL2") ? 0 : 1;   // 2-pixel bilateral kernel
// multi-GPU: run the halo exchange of pass i+1 beside the interior rows of pass i (0 = every
// exchange on the compute stream, between the passes)
static int g_opt_overlap = getenv("WT_NO_OVERLAP") ? 0 : 1;
// compute units (of 256) the interior launch leaves free for the RCCL kernels of that exchange
static int g_opt_overlap_reserve = getenv("WT_OVERLAP_RESERVE") ? atoi(getenv("WT_OVERLAP_RESERVE")) : 16;
// measurement aid: split the passes of a strip plan as the overlapped schedule does, without any
// exchange (FLAG_NO_EXCHANGE runs on one GPU: what do the edge / interior launches cost?)
static int g_opt_split_dry = 0;

static void wt_set_fused64(int on);     // wt_f64.h (included at the end of this file)
static void wt_set_select64_list(int on);
static void wt_set_f64_pairs(int on);
static void wt_set_stencil64(int on);
static void wt_set_hist_window(int on);
// wt_decompose_sum_host: pipeline the PCIe legs with the passes (0: upload, passes, download in turn)
static int g_opt_host_pipeline = getenv("WT_NO_HOST_PIPELINE") ? 0 : 1;

extern "C" int wt_set_option(const char *name, int value)
{
    if (!name) WT_FAIL("wt_set_option: null name");
    if (!strcmp(name, "row_kernel")) { g_opt_row_kernel = value != 0; return 0; }
    if (!strcmp(name, "lattice_kernel")) { g_opt_lattice = value != 0; return 0; }
    if (!strcmp(name, "bilateral2")) { g_opt_bilateral2 = value != 0; return 0; }
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
    return wt_launch_stencil<float, MODE>(stencil_ctx(p), a, s, name);
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

static int maybe_exchange(wt_plan *p, int plane, int64_t rows, int flags)
{
    if (p->nranks == 1 || (flags & 2)) return 0;
    return wt_halo_exchange(p, plane, rows);
}

static inline int64_t scale_halo(const wt_plan *p, int s) { return (int64_t)(family_taps(p->family) / 2) << s; }

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
static int launch_bilateral(wt_plan *p, const float *in, const float *var, float *out, float *out_w, int s,
                            float f1 = 1.f, float f2 = 1.f, int rev = 0)
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
    const bool small = (1 << s) < 4, b3 = p->family == WT_B3SPLINE;
    if (g_opt_bilateral2) {                              // two pixels per thread: 4 waves per SIMD
        WT_TRY(chain_geometry(p, s, a, grid, block, ((p->g.W + 1) / 2 + 63) / 64));
        ProfScope ps(p->ctx, "wt_bilateral2_kernel");
        static const int lds_pad = getenv("WT_BIL_LDS_PAD") ? atoi(getenv("WT_BIL_LDS_PAD")) : 0;   // experiments: dynamic LDS to cap the workgroups per CU
        if (b3) hipLaunchKernelGGL((wt_bilateral2_kernel<5>), grid, block, lds_pad, p->ctx->stream, a);
        else hipLaunchKernelGGL((wt_bilateral2_kernel<3>), grid, block, lds_pad, p->ctx->stream, a);
        WT_HIP(hipGetLastError());
        return 0;
    }
    WT_TRY(chain_geometry(p, s, a, grid, block));
    ProfScope ps(p->ctx, "wt_bilateral_kernel");
    if (b3 && small) hipLaunchKernelGGL((wt_bilateral_kernel<5, true>), grid, block, 0, p->ctx->stream, a);
    else if (b3) hipLaunchKernelGGL((wt_bilateral_kernel<5, false>), grid, block, 0, p->ctx->stream, a);
    else if (small) hipLaunchKernelGGL((wt_bilateral_kernel<3, true>), grid, block, 0, p->ctx->stream, a);
    else hipLaunchKernelGGL((wt_bilateral_kernel<3, false>), grid, block, 0, p->ctx->stream, a);
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
static int g_opt_hist_window = getenv("WT_NO_HIST_WINDOW") ? 0 : 1;
static inline uint32_t *hist_base_word(wt_ctx *c) { return c->d_hist + WT_HIST_BINS + 30; }
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
static int denoise_sum_rows(wt_plan *p, int count, int dst, int n_den, const double *tau, const double *wgt, int soft, int r0, int r1);

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
            WT_TRY(launch_bilateral(p, in, nullptr, oc, ow, s, f1, f2, (flags & 8) != 0));
        }
        if (overlap) WT_HIP(hipEventRecord(p->scale_ev[s], p->ctx->stream));     // w_s is written
        cur = nxt;
    }
    // the per-scale work on w_s that follows (wt_wow_scale, wt_abs_median) may run beside the scales still queued
    p->overlap_scales = overlap ? level : 0;
    p->overlap_ok = overlap;
    return 0;
}

// =============================================================================================
// pointwise ops
// =============================================================================================
extern "C" int wt_plane_sum(wt_plan *p, int first, int count, int dst)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_plane_sum: null plan");
    if (count < 1 || count > WT_MAX_SUM_PLANES) WT_FAIL("wt_plane_sum: count %d out of range [1,%d]", count, WT_MAX_SUM_PLANES);
    if (first < 0 || first + count - 1 > p->max_level) WT_FAIL("wt_plane_sum: planes [%d,%d) outside [0,%d]", first, first + count, p->max_level);
    SumArgs a{};
    a.n = count;
    for (int i = 0; i < count; ++i) {
        float *b = nullptr;
        WT_TRY(plane_base(p, first + i, &b));
        a.p[i] = b;
    }
    float *o = nullptr;
    WT_TRY(plane_base(p, dst, &o));
    const int64_t n4 = plan_n4(p);
    ProfScope ps(p->ctx, "wt_plane_sum_kernel");
    static const int64_t sum_grid = getenv("WT_SUM_GRID") ? atoll(getenv("WT_SUM_GRID")) : ((int64_t)1 << 30);
    const int grid = (int)std::min<int64_t>((n4 + 255) / 256, sum_grid);
    hipLaunchKernelGGL(wt_plane_sum_kernel, dim3(grid), dim3(256), 0, p->ctx->stream, a, o, n4);
    WT_HIP(hipGetLastError());
    return 0;
}

static int noise_ptr(wt_plan *p, int noise_plane, float **np_);

extern "C" int wt_denoise_sum(wt_plan *p, int first, int count, int dst, int n_den, const double *tau,
                              const double *wgt, int soft, int noise_plane, int write_back)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_denoise_sum: null plan");
    if (count < 1 || count > WT_MAX_SUM_PLANES) WT_FAIL("wt_denoise_sum: count %d out of range [1,%d]", count, WT_MAX_SUM_PLANES);
    if (first < 0 || first + count - 1 > p->max_level) WT_FAIL("wt_denoise_sum: planes [%d,%d) outside [0,%d]", first, first + count, p->max_level);
    if (n_den < 0 || n_den > count) WT_FAIL("wt_denoise_sum: n_den %d outside [0,%d]", n_den, count);
    if (n_den > 0 && (!tau || !wgt)) WT_FAIL("wt_denoise_sum: null tau/wgt");
    DenoiseSumArgs a{};
    a.n = count; a.n_den = n_den; a.soft = soft; a.write_back = write_back;
    for (int i = 0; i < count; ++i) {
        float *b = nullptr;
        WT_TRY(plane_base(p, first + i, &b));
        a.p[i] = b;
        a.tau[i] = i < n_den ? tau[i] : 0.0;
        a.wgt[i] = i < n_den ? (float)wgt[i] : 1.f;
    }
    float *o = nullptr, *nz = nullptr;
    WT_TRY(plane_base(p, dst, &o));
    WT_TRY(noise_ptr(p, noise_plane, &nz));
    const int64_t n4 = plan_n4(p);
    ProfScope ps(p->ctx, "wt_denoise_sum_kernel");
    hipLaunchKernelGGL(wt_denoise_sum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, p->ctx->stream, a, nz, o, n4);
    WT_HIP(hipGetLastError());
    return 0;
}

// wt_denoise_sum over the strip-local rows [r0, r1) of planes 0 .. count-1 (planes are contiguous with
// pitch P: a row range is a flat range); planes are not written back
static int denoise_sum_rows(wt_plan *p, int count, int dst, int n_den, const double *tau, const double *wgt, int soft, int r0, int r1)
{
    if (r1 <= r0) return 0;
    DenoiseSumArgs a{};
    a.n = count; a.n_den = n_den; a.soft = soft; a.write_back = 0;
    const size_t off = (size_t)r0 * p->g.P;
    for (int i = 0; i < count; ++i) {
        float *b = nullptr;
        WT_TRY(plane_base(p, i, &b));
        a.p[i] = b + off;
        a.tau[i] = i < n_den ? tau[i] : 0.0;
        a.wgt[i] = i < n_den ? (float)wgt[i] : 1.f;
    }
    float *o = nullptr;
    WT_TRY(plane_base(p, dst, &o));
    const int64_t n4 = (int64_t)(r1 - r0) * p->g.P / 4;
    ProfScope ps(p->ctx, "wt_denoise_sum_kernel");
    hipLaunchKernelGGL(wt_denoise_sum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, p->ctx->stream, a, (const float *)nullptr, o + off, n4);
    WT_HIP(hipGetLastError());
    return 0;
}

static int noise_ptr(wt_plan *p, int noise_plane, float **np_)
{
    *np_ = nullptr;
    if (noise_plane == WT_PLANE_NONE) return 0;
    return plane_base(p, noise_plane, np_);
}

extern "C" int wt_significance(wt_plan *p, int plane, int dst, double tau, int soft, int noise_plane)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_significance: null plan");
    if (!(tau > 0.0)) WT_FAIL("wt_significance: tau must be positive (the sigma==0 / noise==0 short-circuits of wavelets.py:130-143 are host-side)");
    float *c = nullptr, *d = nullptr, *nz = nullptr;
    WT_TRY(plane_base(p, plane, &c));
    WT_TRY(plane_base(p, dst, &d));
    WT_TRY(noise_ptr(p, noise_plane, &nz));
    const int64_t n4 = plan_n4(p);
    ProfScope ps(p->ctx, "wt_signif_kernel");
    hipLaunchKernelGGL(wt_signif_kernel, dim3(flat_grid(n4)), dim3(256), 0, p->ctx->stream, c, nz, d, n4, tau, 1.f, soft, 0);
    WT_HIP(hipGetLastError());
    return 0;
}

extern "C" int wt_denoise(wt_plan *p, int plane, double tau, double wgt, int soft, int noise_plane)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_denoise: null plan");
    if (!(tau > 0.0)) WT_FAIL("wt_denoise: tau must be positive");
    float *c = nullptr, *nz = nullptr;
    WT_TRY(plane_base(p, plane, &c));
    WT_TRY(noise_ptr(p, noise_plane, &nz));
    const int64_t n4 = plan_n4(p);
    ProfScope ps(p->ctx, "wt_signif_kernel");
    hipLaunchKernelGGL(wt_signif_kernel, dim3(flat_grid(n4)), dim3(256), 0, p->ctx->stream, c, nz, c, n4, tau, (float)wgt, soft, 1);
    WT_HIP(hipGetLastError());
    return 0;
}

extern "C" int wt_wow_update(wt_plan *p, int plane, int power_plane, double tau, int soft, int noise_plane, float factor, int gamma_plane)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_wow_update: null plan");
    float *c = nullptr, *pw = nullptr, *nz = nullptr, *gm = nullptr;
    WT_TRY(plane_base(p, plane, &c));
    if (power_plane != WT_PLANE_NONE) WT_TRY(plane_base(p, power_plane, &pw));
    if (gamma_plane != WT_PLANE_NONE) WT_TRY(plane_base(p, gamma_plane, &gm));
    WT_TRY(noise_ptr(p, noise_plane, &nz));
    const int64_t n4 = plan_n4(p);
    ProfScope ps(p->ctx, "wt_wow_kernel");
    hipLaunchKernelGGL(wt_wow_kernel, dim3(flat_grid(n4)), dim3(256), 0, p->ctx->stream, c, pw, nz, gm, n4, tau, soft, factor);
    WT_HIP(hipGetLastError());
    return 0;
}

// Fused wow per-scale update: local power conv_s(c^2) (watroo/utils.py:194) is formed inside
// the kernel that applies wt_wow_update's pointwise step, and the result is written to a spare
// plane whose pointer is then swapped with the coefficient plane ("in place" at pointer level:
// the neighbours' taps still need the old values while the kernel runs).
extern "C" int wt_wow_scale(wt_plan *p, int plane, int s, double tau, int soft, int noise_plane,
                            float factor, int gamma_plane, int flags)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_wow_scale: null plan");
    if (plane < 0 || plane > p->max_level) WT_FAIL("wt_wow_scale: plane %d is not a coefficient plane", plane);
    WT_TRY(check_scale(p, s, "wt_wow_scale"));
    const int spare = WT_PLANE_SCRATCH(3);
    // Right behind a bilateral transform the update of w_s only needs scale s of it (its event): it runs on the
    // side stream, beside the bilateral kernels of the later scales (plain mode: no maps to order against)
    const bool side = g_opt_wow_overlap && p->overlap_ok && plane < p->overlap_scales && noise_plane == WT_PLANE_NONE &&
                      gamma_plane == WT_PLANE_NONE && p->nranks == 1 && !p->ntaps;
    WtSideScope side_scope(p->ctx, side ? p->scale_ev[plane] : nullptr, side);
    if (!side_scope.ok()) return 2;
    float *c = nullptr, *t = nullptr, *nz = nullptr, *gm = nullptr;
    WT_TRY(plane_base(p, plane, &c));
    WT_TRY(plane_base(p, spare, &t));
    WT_TRY(noise_ptr(p, noise_plane, &nz));
    if (gamma_plane != WT_PLANE_NONE) WT_TRY(plane_base(p, gamma_plane, &gm));
    WT_TRY(maybe_exchange(p, plane, scale_halo(p, s), flags));
    ChainArgs a{};
    a.in = c; a.out_c = t; a.out_w = nullptr; a.aux = nullptr;
    a.noise = nz; a.gamma = gm; a.tau = tau; a.factor = factor; a.soft = soft; a.whiten = 1;
    // (no per-pixel noise map, no gamma accumulator: the instantiation without conditional loads)
    if (!nz && !gm) WT_TRY(launch_chain_args<MODE_WOW_PLAIN>(p, a, s, "wt_chain_kernel<wow>"));
    else if (!nz) WT_TRY(launch_chain_args<MODE_WOW_GAMMA>(p, a, s, "wt_chain_kernel<wow>"));
    else WT_TRY(launch_chain_args<MODE_WOW>(p, a, s, "wt_chain_kernel<wow>"));
    std::swap(p->coef[plane], p->scratch[3]);     // both are "first margin row" pointers
    return 0;
}

extern "C" int wt_gamma_blend(wt_plan *p, int recon, int gamma_plane, float gmin, float gmax, float inv_gamma, float h)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_gamma_blend: null plan");
    float *r = nullptr, *g = nullptr;
    WT_TRY(plane_base(p, recon, &r));
    WT_TRY(plane_base(p, gamma_plane, &g));
    const int64_t n4 = plan_n4(p);
    ProfScope ps(p->ctx, "wt_gamma_kernel");
    hipLaunchKernelGGL(wt_gamma_kernel, dim3(flat_grid(n4)), dim3(256), 0, p->ctx->stream, r, g, n4, gmin, gmax - gmin, inv_gamma, h);
    WT_HIP(hipGetLastError());
    return 0;
}

extern "C" int wt_anscombe(wt_plan *p, int src, int dst, float alpha, float g, float sigma, int inverse)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_anscombe: null plan");
    if (alpha == 0.f) WT_FAIL("wt_anscombe: alpha must be non-zero");
    float *s = nullptr, *d = nullptr;
    WT_TRY(plane_base(p, src, &s));
    WT_TRY(plane_base(p, dst, &d));
    // scalar terms are formed in double like the python floats of wavelets.py:17,19
    const double a = alpha, gg = g, sg = sigma;
    float c1, c2, c3;
    if (inverse) { c1 = (float)(a * gg); c2 = (float)(sg * sg); c3 = (float)(3.0 * a / 8.0); }
    else { c1 = (float)(3.0 * a * a / 8.0); c2 = (float)(sg * sg); c3 = (float)(a * gg); }
    const int64_t n4 = plan_n4(p);
    ProfScope ps(p->ctx, "wt_anscombe_kernel");
    hipLaunchKernelGGL(wt_anscombe_kernel, dim3(flat_grid(n4)), dim3(256), 0, p->ctx->stream, s, d, n4, alpha, c1, c2, c3, inverse);
    WT_HIP(hipGetLastError());
    return 0;
}

// =============================================================================================
// 3-D cubes: a (Z, Y, X) cube is a (Z*Y) x X image on the plan
// =============================================================================================
extern "C" int wt_binary(wt_plan *p, int op, int a, int b, int dst);
static int conv3d_planes(wt_plan *p, float *in, float *tmp, float *out, int s, int depth)
{
    const Geo whole = p->g;
    const int Y = whole.H / depth;
    if (p->ntaps) {      // user-defined taps: rows -> scratch 15, axis 1 -> scratch 12, axis 0 -> out
        float *t2 = nullptr;
        WT_TRY(plane_base(p, WT_PLANE_SCRATCH(12), &t2));
        if (in == t2 || out == t2 || tmp == t2) WT_FAIL("3-D filter: scratch plane 12 is used internally for user-defined taps");
        const CustomTaps t = plan_taps(p);
        dim3 grid((whole.W + 255) / 256, (unsigned)std::min(whole.H, 32768)), block(256);
        ProfScope ps(p->ctx, "wt_custom_kernels");
        hipLaunchKernelGGL(wt_custom_rows_kernel, grid, block, 0, p->ctx->stream, (const float *)in, tmp, whole, 1 << s, t, 0);
        hipLaunchKernelGGL(wt_custom_axis_kernel, grid, block, 0, p->ctx->stream, (const float *)tmp, t2, whole.W, whole.P, Y, depth,
                           1 << s, whole.border, t, 1);
        hipLaunchKernelGGL(wt_custom_axis_kernel, grid, block, 0, p->ctx->stream, (const float *)t2, out, whole.W, whole.P, Y, depth,
                           1 << s, whole.border, t, 0);
        WT_HIP(hipGetLastError());
        return 0;
    }
    // per-slice 2-D filter: the single-scale kernels run on each Y x X slice as its own image
    p->g.H = Y;
    p->g.nrows = Y;
    int rc = 0;
    for (int z = 0; z < depth && !rc; ++z) {
        const size_t off = (size_t)z * Y * whole.P;
        rc = launch_chain<MODE_SMOOTH>(p, in + off, tmp + off, nullptr, s, 1.f, 1.f, 0, "wt_chain_kernel<smooth>");
    }
    p->g = whole;
    if (rc) return rc;
    const int64_t n4 = plan_n4(p);
    ProfScope ps(p->ctx, "wt_zfilter_kernel");
    if (p->family == WT_B3SPLINE)
        hipLaunchKernelGGL((wt_zfilter_kernel<5>), dim3(flat_grid(n4)), dim3(256), 0, p->ctx->stream, tmp, out, n4, whole.P / 4, Y, depth, 1 << s, whole.border);
    else
        hipLaunchKernelGGL((wt_zfilter_kernel<3>), dim3(flat_grid(n4)), dim3(256), 0, p->ctx->stream, tmp, out, n4, whole.P / 4, Y, depth, 1 << s, whole.border);
    WT_HIP(hipGetLastError());
    return 0;
}

static int check3d(const wt_plan *p, int depth, int s, const char *who)
{
    if (p->nranks != 1 || (p->g.border != 0 && p->g.border != 1))
        WT_FAIL("%s: single-GPU plans with the symmetric border (whole cube or polyphase) only", who);
    if (depth < 1 || p->g.H % depth) WT_FAIL("%s: plan height %d is not a multiple of depth %d", who, p->g.H, depth);
    if (s < 0 || s > 20) WT_FAIL("%s: scale %d out of range", who, s);
    return 0;
}

extern "C" int wt_smooth3d(wt_plan *p, int src, int dst, int s, int depth)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_smooth3d: null plan");
    WT_TRY(check3d(p, depth, s, "wt_smooth3d"));
    const int tmpid = WT_PLANE_SCRATCH(15);
    if (src == dst || src == tmpid || dst == tmpid) WT_FAIL("wt_smooth3d: src, dst and scratch 15 must differ");
    float *in = nullptr, *tmp = nullptr, *out = nullptr;
    WT_TRY(plane_base(p, src, &in));
    WT_TRY(plane_base(p, tmpid, &tmp));
    WT_TRY(plane_base(p, dst, &out));
    return conv3d_planes(p, in, tmp, out, s, depth);
}

// sdev_loc(..., variance=True) * f1 * f2 of a cube (watroo/wavelets.py:24-32, :434-436) into `dst`
extern "C" int wt_local_variance3d(wt_plan *p, int src, int dst, int s, int depth, float f1, float f2)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_local_variance3d: null plan");
    WT_TRY(check3d(p, depth, s, "wt_local_variance3d"));
    const int tmpid = WT_PLANE_SCRATCH(15), sqid = WT_PLANE_SCRATCH(14), mid = WT_PLANE_SCRATCH(13);
    if (src == dst || src == tmpid || src == sqid || src == mid || dst == tmpid || dst == sqid || dst == mid)
        WT_FAIL("wt_local_variance3d: src / dst must differ from each other and from scratch 13-15");
    float *in = nullptr, *tmp = nullptr, *sq = nullptr, *mean = nullptr, *out = nullptr;
    WT_TRY(plane_base(p, src, &in));
    WT_TRY(plane_base(p, tmpid, &tmp));
    WT_TRY(plane_base(p, sqid, &sq));
    WT_TRY(plane_base(p, mid, &mean));
    WT_TRY(plane_base(p, dst, &out));
    WT_TRY(conv3d_planes(p, in, tmp, mean, s, depth));          // conv(I)
    WT_TRY(wt_binary(p, WT_OP_MUL, src, src, sqid));            // I^2
    WT_TRY(conv3d_planes(p, sq, tmp, out, s, depth));           // conv(I^2)
    const int64_t n4 = plan_n4(p);
    ProfScope ps(p->ctx, "wt_var_moments_kernel");
    hipLaunchKernelGGL(wt_var_moments_kernel, dim3(flat_grid(n4)), dim3(256), 0, p->ctx->stream, (const float *)mean,
                       (const float *)out, out, n4, f1, f2);
    WT_HIP(hipGetLastError());
    return 0;
}

// atrous_convolution(cube, 3-D kernel, bilateral_variance=var, s) - watroo/wavelets.py:74-105
extern "C" int wt_bilateral3d_conv(wt_plan *p, int src, int var, int dst, int s, int depth)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_bilateral3d_conv: null plan");
    WT_TRY(check3d(p, depth, s, "wt_bilateral3d_conv"));
    if (src == dst || var == dst) WT_FAIL("wt_bilateral3d_conv: dst must differ from src and var");
    float *in = nullptr, *v = nullptr, *out = nullptr;
    WT_TRY(plane_base(p, src, &in));
    WT_TRY(plane_base(p, var, &v));
    WT_TRY(plane_base(p, dst, &out));
    const int Y = p->g.H / depth;
    dim3 grid((p->g.W + 255) / 256, (unsigned)std::min(p->g.H, 32768)), block(256);
    if (p->ntaps) {
        ProfScope ps(p->ctx, "wt_bilateral_custom_kernel");
        hipLaunchKernelGGL(wt_bilateral_custom_kernel, grid, block, 0, p->ctx->stream, (const float *)in, (const float *)v, out,
                           (float *)nullptr, p->g, Y, depth, 1 << s, plan_taps(p), 0);
        WT_HIP(hipGetLastError());
        return 0;
    }
    ProfScope ps(p->ctx, "wt_bilateral3d_kernel");
    if (p->family == WT_B3SPLINE)
        hipLaunchKernelGGL((wt_bilateral3d_kernel<5>), grid, block, 0, p->ctx->stream, (const float *)in, (const float *)v, out,
                           p->g.W, p->g.P, Y, depth, 1 << s, p->g.border);
    else
        hipLaunchKernelGGL((wt_bilateral3d_kernel<3>), grid, block, 0, p->ctx->stream, (const float *)in, (const float *)v, out,
                           p->g.W, p->g.P, Y, depth, 1 << s, p->g.border);
    WT_HIP(hipGetLastError());
    return 0;
}

extern "C" int wt_decompose3d(wt_plan *p, int src, int level, int depth)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_decompose3d: null plan");
    WT_TRY(check3d(p, depth, 0, "wt_decompose3d"));
    if (level < 0 || level > p->max_level) WT_FAIL("wt_decompose3d: level %d exceeds plan max_level %d", level, p->max_level);
    if (src >= 0 && src <= level) WT_FAIL("wt_decompose3d: src plane %d is one of the output planes", src);
    if (level == 0) return wt_copy_plane(p, src, 0);
    int cur = src;
    for (int s = 0; s < level; ++s) {
        const int nxt = (s == level - 1) ? level : WT_PLANE_SCRATCH(s & 1);
        if (cur == nxt) WT_FAIL("wt_decompose3d: scratch planes 0/1 are used internally");
        float *in = nullptr, *tmp = nullptr, *oc = nullptr;
        WT_TRY(plane_base(p, cur, &in));
        WT_TRY(plane_base(p, WT_PLANE_SCRATCH(15), &tmp));
        WT_TRY(plane_base(p, nxt, &oc));
        WT_TRY(conv3d_planes(p, in, tmp, oc, s, depth));
        WT_TRY(wt_binary(p, WT_OP_SUB, cur, nxt, s));        // w_s = c_s - c_{s+1}   (wavelets.py:442)
        cur = nxt;
    }
    return 0;
}

// =============================================================================================
// Richardson-Lucy support
// =============================================================================================
extern "C" int wt_filter2d_ex(wt_plan *p, int src, int dst, const float *kernel, int kh, int kw, int ay, int ax,
                             int border, int flags)
{
    WtGuard guard_(ctx_of(p));
    if (!p || !kernel) WT_FAIL("wt_filter2d: null pointer");
    if (kh < 1 || kw < 1 || (int64_t)kh * kw > (1 << 22)) WT_FAIL("wt_filter2d: kernel %d x %d unsupported (up to 2^22 taps)", kh, kw);
    if (ay < 0 || ay >= kh || ax < 0 || ax >= kw) WT_FAIL("wt_filter2d: anchor (%d, %d) outside the %d x %d kernel", ay, ax, kh, kw);
    if (src == dst) WT_FAIL("wt_filter2d: src and dst must differ");
    if (border != WT_BORDER_SYMMETRIC && border != WT_BORDER_PERIODIC) WT_FAIL("wt_filter2d: border %d unsupported (symmetric or periodic)", border);
    if (p->g.border) WT_FAIL("wt_filter2d ignores the plan's border mode; reset it to symmetric first");
    const bool wrap = border == WT_BORDER_PERIODIC;
    if (wrap && (p->nranks > 1 || p->g.row0 != 0 || p->g.nrows != p->g.H))
        WT_FAIL("wt_filter2d: the periodic border needs a whole-image plan");
    const int reach = std::max(ay, kh - 1 - ay);
    if (p->nranks > 1 && reach > p->g.halo) WT_FAIL("wt_filter2d: kernel needs %d halo rows, plan has %d", reach, p->g.halo);
    wt_ctx *c = p->ctx;
    float *in = nullptr, *o = nullptr;
    WT_TRY(plane_base(p, src, &in));
    WT_TRY(plane_base(p, dst, &o));
    WT_TRY(maybe_exchange(p, src, reach, flags));
    const size_t ntaps = (size_t)kh * kw;
    // the taps come from caller-owned memory: drain the stream (the previous PSF may still be read),
    // copy synchronously (small PSFs through the pinned scratch, as before)
    WT_HIP(hipStreamSynchronize(c->stream));
    if (ntaps > c->d_psf_cap) {
        (void)hipFree(c->d_psf);
        c->d_psf = nullptr;
        c->d_psf_cap = 0;
        WT_HIP(hipMalloc(&c->d_psf, ntaps * sizeof(float)));
        c->d_psf_cap = ntaps;
    }
    if (ntaps * sizeof(float) <= 65536) {
        memcpy(c->h_pinned, kernel, ntaps * sizeof(float));
        WT_HIP(hipMemcpyAsync(c->d_psf, c->h_pinned, ntaps * sizeof(float), hipMemcpyHostToDevice, c->stream));
    } else {
        WT_HIP(hipMemcpy(c->d_psf, kernel, ntaps * sizeof(float), hipMemcpyHostToDevice));
    }
    dim3 grid((p->g.W + WT_F2D_TW - 1) / WT_F2D_TW, (p->g.nrows + WT_F2D_TH - 1) / WT_F2D_TH), block(64, 4);
    if (grid.y > 65535u) WT_FAIL("wt_filter2d: strip too tall");
    // Bands (round 3: the reference has no PSF size limit, watroo/utils.py:245-257): a launch takes a
    // window of at most 4096 taps whose LDS tile fits 96 KB; the windows tile the PSF and every launch
    // after the first accumulates.  A PSF that fits is one launch, as before.
    const int bw = std::min(kw, 512);
    int bh = std::max(1, std::min(kh, 4096 / bw));
    while (bh > 1 && (size_t)(WT_F2D_TW + bw - 1) * (WT_F2D_TH + bh - 1) * sizeof(float) > 96 * 1024) --bh;
    const size_t lds = (size_t)(WT_F2D_TW + bw - 1) * (WT_F2D_TH + bh - 1) * sizeof(float);
    ProfScope ps(c, "wt_filter2d_kernel");
    bool first = true;
    for (int i0 = 0; i0 < kh; i0 += bh) {
        for (int j0 = 0; j0 < kw; j0 += bw) {
            const int h = std::min(bh, kh - i0), w = std::min(bw, kw - j0);
            const float *sub = c->d_psf + (size_t)i0 * kw + j0;
            #define WT_F2D_LAUNCH(WRAP, ACC)                                                                                        \
                do {                                                                                                                 \
                    if (lds > 64 * 1024)                                                                                             \
                        WT_HIP(hipFuncSetAttribute((const void *)wt_filter2d_kernel<WRAP, ACC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                    hipLaunchKernelGGL((wt_filter2d_kernel<WRAP, ACC>), grid, block, lds, c->stream, (const float *)in, o, p->g, sub, kw, h, w, ay - i0, ax - j0); \
                } while (0)
            if (wrap) { if (first) WT_F2D_LAUNCH(true, false); else WT_F2D_LAUNCH(true, true); }
            else { if (first) WT_F2D_LAUNCH(false, false); else WT_F2D_LAUNCH(false, true); }
            #undef WT_F2D_LAUNCH
            first = false;
        }
    }
    WT_HIP(hipGetLastError());
    return 0;
}

// ---- circular products through the FFT (wt_fft.h; watroo/utils.py:245-254, 284)
extern "C" int wt_fft_supported(int64_t H, int64_t W, int *ok)
{
    if (!ok) WT_FAIL("wt_fft_supported: null pointer");
    *ok = (H <= WT_FFT_MAX_N && W <= WT_FFT_MAX_N && wt_fft_size_ok((int)H, (int)W)) ? 1 : 0;
    return 0;
}

static int fft_plan_check(const wt_plan *p, const char *who)
{
    if (p->nranks != 1 || p->g.row0 != 0 || p->g.nrows != p->g.H) WT_FAIL("%s: whole-image plans only", who);
    if (!wt_fft_size_ok(p->g.H, p->g.W)) WT_FAIL("%s: image %d x %d is not a power of two per side (2 .. %d)", who, p->g.H, p->g.W, WT_FFT_MAX_N);
    return 0;
}

extern "C" int wt_fft_spectrum(wt_plan *p, int src)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_fft_spectrum: null plan");
    WT_TRY(fft_plan_check(p, "wt_fft_spectrum"));
    float *s = nullptr;
    WT_TRY(plane_base(p, src, &s));
    const size_t before = p->raw_allocs.size();
    WT_TRY(wt_fft_prepare<float>(p->ctx, p->fft, p->g.H, p->g.W, p->raw_allocs));
    if (p->raw_allocs.size() != before) p->raw_bytes += (size_t)3 * p->g.H * p->g.W * sizeof(float2) + (size_t)(p->g.H + p->g.W) / 2 * sizeof(float2);
    return wt_fft_set_spectrum<float>(p->ctx, p->fft, s, p->g.P);
}

extern "C" int wt_fft_apply(wt_plan *p, int src, int dst, int conj)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_fft_apply: null plan");
    WT_TRY(fft_plan_check(p, "wt_fft_apply"));
    float *s = nullptr, *d = nullptr;
    WT_TRY(plane_base(p, src, &s));
    WT_TRY(plane_base(p, dst, &d));
    return wt_fft_apply_t<float>(p->ctx, p->fft, s, d, p->g.P, conj);
}

extern "C" int wt_filter2d(wt_plan *p, int src, int dst, const float *kernel, int kh, int kw, int flags)
{
    WtGuard guard_(ctx_of(p));
    return wt_filter2d_ex(p, src, dst, kernel, kh, kw, kh / 2, kw / 2, WT_BORDER_SYMMETRIC, flags);
}

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

extern "C" int wt_taps_conv(wt_plan *p, int src, int var, int dst, const int32_t *offsets, const float *weights, int ntaps,
                            float center_weight, int has_center, int depth, int pad_mode, float fill_value)
{
    WtGuard guard_(ctx_of(p));
    if (pad_mode > WT_PAD_CONSTANT) WT_FAIL("wt_taps_conv: unknown pad mode %d (the polyphase modes take a dilation: wt_taps_conv_ex)", pad_mode);
    return wt_taps_conv_ex(p, src, var, dst, offsets, weights, ntaps, center_weight, has_center, depth, pad_mode, fill_value, 1);
}

extern "C" int wt_variance_from_moments(wt_plan *p, int mean, int meansq, int dst, float f1, float f2, int take_sqrt)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_variance_from_moments: null plan");
    float *m = nullptr, *q = nullptr, *d = nullptr;
    WT_TRY(plane_base(p, mean, &m));
    WT_TRY(plane_base(p, meansq, &q));
    WT_TRY(plane_base(p, dst, &d));
    const int64_t n4 = plan_n4(p);
    ProfScope ps(p->ctx, "wt_var_moments_kernel");
    hipLaunchKernelGGL(wt_var_moments_kernel, dim3(flat_grid(n4)), dim3(256), 0, p->ctx->stream, (const float *)m, (const float *)q, d, n4, f1, f2,
                       take_sqrt);
    WT_HIP(hipGetLastError());
    return 0;
}

extern "C" int wt_taps_conv_ex(wt_plan *p, int src, int var, int dst, const int32_t *offsets, const float *weights, int ntaps,
                               float center_weight, int has_center, int depth, int pad_mode, float fill_value, int dilation)
{
    WtGuard guard_(ctx_of(p));
    if (!p || (ntaps > 0 && (!offsets || !weights))) WT_FAIL("wt_taps_conv: null pointer");
    if (dilation < 1) WT_FAIL("wt_taps_conv: dilation %d must be positive", dilation);
    if (p->nranks > 1) WT_FAIL("wt_taps_conv: the generic operator is single-GPU (whole images)");
    if (src == dst || var == dst) WT_FAIL("wt_taps_conv: dst must differ from src and var");
    if (pad_mode < WT_PAD_SYMMETRIC || pad_mode > WT_PAD_POLY_MIRROR) WT_FAIL("wt_taps_conv: unknown pad mode %d", pad_mode);
    if (depth < 0 || (depth > 0 && p->g.nrows % depth)) WT_FAIL("wt_taps_conv: %d rows are not a multiple of depth %d", p->g.nrows, depth);
    float *in = nullptr, *o = nullptr, *v = nullptr;
    WT_TRY(plane_base(p, src, &in));
    WT_TRY(plane_base(p, dst, &o));
    if (var != WT_PLANE_NONE) WT_TRY(plane_base(p, var, &v));
    const int32_t *d_offs = nullptr;
    const float *d_wts = nullptr;
    WT_TRY(upload_taplist<float>(p->ctx, offsets, weights, ntaps, &d_offs, &d_wts));
    const int Z = depth > 0 ? depth : 1, Y = p->g.nrows / Z;
    ProfScope ps(p->ctx, "wt_taps_kernel");
    hipLaunchKernelGGL(wt_taps_kernel<float>, dim3((p->g.W + 255) / 256, (unsigned)std::min(p->g.nrows, 32768)), dim3(256), 0, p->ctx->stream,
                       (const float *)in, (const float *)v, o, p->g.W, p->g.P, Y, Z, d_offs, d_wts, ntaps, center_weight, has_center, pad_mode,
                       fill_value, dilation);
    WT_HIP(hipGetLastError());
    return 0;
}

int g_opt_axis_filter = getenv("WT_NO_AXIS_FILTER") ? 0 : 1;

/* K-tap filter along ONE axis of plane src -> dst: out[i] = sum_j weights[j] * in[pad(i + offsets[j])] along axis
 * 2 (x), 1 (y, inside every slice of a cube) or 0 (z, across the `depth` slices); border rule pad_mode / fill /
 * dilation as wt_taps_conv_ex.  The separable form of the scaling functions the tuned kernels do not take
 * (watroo/wavelets.py:152-197: any coefficients_1d) on the tiled kernels of wt_axis.h; tap sets they do not
 * take (more than 33 taps, irregular offsets along y / z, an x reach beyond 2048 pixels) run on the tap-list
 * operator - same bits either way. */
extern "C" int wt_axis_filter(wt_plan *p, int src, int dst, int axis, const int32_t *offsets, const float *weights, int ntaps, int depth,
                              int pad_mode, float fill_value, int dilation)
{
    WtGuard guard_(ctx_of(p));
    if (!p || !offsets || !weights) WT_FAIL("wt_axis_filter: null pointer");
    if (ntaps < 1 || ntaps > 4096) WT_FAIL("wt_axis_filter: %d taps unsupported", ntaps);
    if (axis < 0 || axis > 2) WT_FAIL("wt_axis_filter: axis %d (2 = x, 1 = y, 0 = z)", axis);
    if (dilation < 1) WT_FAIL("wt_axis_filter: dilation %d must be positive", dilation);
    if (p->nranks > 1) WT_FAIL("wt_axis_filter: the generic operator is single-GPU (whole images)");
    if (src == dst) WT_FAIL("wt_axis_filter: dst must differ from src");
    if (pad_mode < WT_PAD_SYMMETRIC || pad_mode > WT_PAD_POLY_MIRROR) WT_FAIL("wt_axis_filter: unknown pad mode %d", pad_mode);
    if (depth < 0 || (depth > 0 && p->g.nrows % depth)) WT_FAIL("wt_axis_filter: %d rows are not a multiple of depth %d", p->g.nrows, depth);
    if (axis == 0 && depth == 0) WT_FAIL("wt_axis_filter: axis 0 needs a cube (depth > 0)");
    float *in = nullptr, *o = nullptr;
    WT_TRY(plane_base(p, src, &in));
    WT_TRY(plane_base(p, dst, &o));
    const int rc = wt_axis_filter_launch<float>(p->ctx, in, o, p->g.W, p->g.P, p->g.nrows, depth, axis, offsets, weights, ntaps, pad_mode,
                                                fill_value, dilation);
    if (rc >= 0) return rc;
    std::vector<int32_t> o3((size_t)ntaps * 3, 0);
    for (int j = 0; j < ntaps; ++j) o3[(size_t)3 * j + axis] = offsets[j];
    return wt_taps_conv_ex(p, src, WT_PLANE_NONE, dst, o3.data(), weights, ntaps, 0.f, 0, depth, pad_mode, fill_value, dilation);
}

extern "C" int wt_binary(wt_plan *p, int op, int a, int b, int dst)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_binary: null plan");
    if (op < 0 || op > WT_OP_ADD_DIV) WT_FAIL("wt_binary: unknown op %d", op);
    float *pa = nullptr, *pb = nullptr, *pd = nullptr;
    WT_TRY(plane_base(p, a, &pa));
    WT_TRY(plane_base(p, b, &pb));
    WT_TRY(plane_base(p, dst, &pd));
    const int64_t n4 = plan_n4(p);
    ProfScope ps(p->ctx, "wt_binary_kernel");
    hipLaunchKernelGGL(wt_binary_kernel, dim3(flat_grid(n4)), dim3(256), 0, p->ctx->stream, pa, pb, pd, n4, op);
    WT_HIP(hipGetLastError());
    return 0;
}

extern "C" int wt_mrs_update(wt_plan *p, int plane, int mrs_plane, double tau, int soft, int noise_plane,
                             int persistent, float inv_pow)
{
    WtGuard guard_(ctx_of(p));
    if (!p) WT_FAIL("wt_mrs_update: null plan");
    if (plane == mrs_plane) WT_FAIL("wt_mrs_update: plane and mrs_plane must differ");
    float *c = nullptr, *m = nullptr, *nz = nullptr;
    WT_TRY(plane_base(p, plane, &c));
    WT_TRY(plane_base(p, mrs_plane, &m));
    WT_TRY(noise_ptr(p, noise_plane, &nz));
    const int64_t n4 = plan_n4(p);
    ProfScope ps(p->ctx, "wt_mrs_kernel");
    hipLaunchKernelGGL(wt_mrs_kernel, dim3(flat_grid(n4)), dim3(256), 0, p->ctx->stream, c, m, nz, n4, tau, soft, persistent, inv_pow);
    WT_HIP(hipGetLastError());
    return 0;
}

// =============================================================================================
// reductions / selection (host-synchronous: they return values)
// =============================================================================================
extern "C" int wt_reduce(wt_plan *p, int plane, double out[4])
{
    WtGuard guard_(ctx_of(p));
    if (!p || !out) WT_FAIL("wt_reduce: null pointer");
    wt_ctx *c = p->ctx;
    float *b = nullptr;
    WT_TRY(plane_base(p, plane, &b));
    // (work items: (row, chunk of 4096 pixels) pairs)
    const int blocks = (int)std::min<int64_t>((int64_t)p->g.nrows * ((p->g.W + 4095) / 4096), c->partial_blocks);
    double *dout = c->d_partials + (size_t)c->partial_blocks * 4;
    {
        ProfScope ps(c, "wt_reduce_kernel");
        hipLaunchKernelGGL(wt_reduce_kernel, dim3(blocks), dim3(256), 0, c->stream, b, p->g.nrows, p->g.P / 4, p->g.W, c->d_partials);
        hipLaunchKernelGGL(wt_reduce_final_kernel, dim3(1), dim3(256), 0, c->stream, c->d_partials, blocks, dout);
    }
    WT_HIP(hipGetLastError());
    if (p->nranks > 1) {
        if (!c->comm) WT_FAIL("wt_reduce: multi-rank plan without communicator");
        WT_NCCL(g_rccl.AllReduce(dout, dout, 2, NCCL_FLOAT64, NCCL_SUM, c->comm, c->stream));
        WT_NCCL(g_rccl.AllReduce(dout + 2, dout + 2, 1, NCCL_FLOAT64, NCCL_MIN, c->comm, c->stream));
        WT_NCCL(g_rccl.AllReduce(dout + 3, dout + 3, 1, NCCL_FLOAT64, NCCL_MAX, c->comm, c->stream));
    }
    WT_HIP(hipMemcpyAsync(c->h_pinned, dout, 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    WT_HIP(hipStreamSynchronize(c->stream));
    memcpy(out, c->h_pinned, 4 * sizeof(double));
    return 0;
}

// One histogram pass of the radix select + the device-side step that folds the selected bin into
// the state (no host round trip: the three passes chain on the stream).
static int select_pass(wt_plan *p, const float *b, WtSelectState *st, uint32_t prefix_mask, int shift, uint32_t bin_mask, int last,
                       bool have_hist = false)
{
    wt_ctx *c = p->ctx;
    if (!have_hist) {
        ProfScope ps(c, "wt_hist_kernel");
        // first level (no prefix yet): every element is binned - four interleaved LDS copies;
        // 4 blocks per CU (32 KB of LDS each) against 8 for the later levels
        const int X4 = (p->g.W + 3) / 4, nchunk = (X4 + 256 * WT_HIST_UNROLL - 1) / (256 * WT_HIST_UNROLL);
        const int64_t nitems = (int64_t)p->g.nrows * nchunk;
        if (prefix_mask == 0u)
            hipLaunchKernelGGL(wt_hist_kernel<4>, dim3((unsigned)std::min<int64_t>(nitems, 4 * c->num_cus)), dim3(256), 0, c->stream, b,
                               p->g.nrows, p->g.P / 4, p->g.W, prefix_mask, (const WtSelectState *)st, shift, bin_mask, c->d_hist);
        else
            hipLaunchKernelGGL(wt_hist_kernel<1>, dim3((unsigned)std::min<int64_t>(nitems, 8 * c->num_cus)), dim3(256), 0, c->stream, b,
                               p->g.nrows, p->g.P / 4, p->g.W, prefix_mask, (const WtSelectState *)st, shift, bin_mask, c->d_hist);
    }
    WT_HIP(hipGetLastError());
    if (p->nranks > 1) {
        if (!c->comm) WT_FAIL("wt_abs_median: multi-rank plan without communicator");
        WT_NCCL(g_rccl.AllReduce(c->d_hist, c->d_hist, WT_HIST_BINS, NCCL_UINT32, NCCL_SUM, c->comm, c->stream));
    }
    hipLaunchKernelGGL(wt_select_step_kernel, dim3(1), dim3(256), 0, c->stream, c->d_hist, st, (int)bin_mask + 1, shift, last);
    WT_HIP(hipGetLastError());
    return 0;
}

extern "C" int wt_abs_median(wt_plan *p, int plane, float *median)
{
    WtGuard guard_(ctx_of(p));
    if (!p || !median) WT_FAIL("wt_abs_median: null pointer");
    wt_ctx *c = p->ctx;
    // MAD of a detail plane right behind a bilateral transform: beside the scales still queued (side stream)
    const bool side = g_opt_wow_overlap && p->overlap_ok && plane >= 0 && plane < p->overlap_scales && p->nranks == 1;
    WtSideScope side_scope(c, side ? p->scale_ev[plane] : nullptr, side);
    if (!side_scope.ok()) return 2;
    // a fused pass has histogrammed the first level of this plane (flag bit4 of wt_decompose /
    // wt_decompose_pass) and nothing has touched the plane or the bins since: one pass less over it
    const bool pre = c->prehist_plan == p && c->prehist_plane == plane;
    c->prehist_plan = nullptr;
    float *b = nullptr;
    WT_TRY(plane_base(p, plane, &b));
    const int64_t N = (int64_t)p->g.H * p->g.W;   // global element count
    const int64_t klo = (N - 1) / 2;
    // state on the device (behind the histogram and the upper-median word); initialised from pinned memory
    WtSelectState *st = (WtSelectState *)(c->d_hist + WT_HIST_BINS + 4);
    WtSelectState *hst = (WtSelectState *)c->h_pinned;
    hst->k = (unsigned long long)klo; hst->cum_le = 0; hst->prefix = 0; hst->failed = 0;
    const bool windowed = pre && c->prehist_windowed;
    c->prehist_windowed = false;
    WtSelectState res_st{};
    auto run = [&](bool have_hist, bool window) -> int {
        WT_HIP(hipMemcpyAsync(st, hst, sizeof(WtSelectState), hipMemcpyHostToDevice, c->stream));
        if (!have_hist) WT_HIP(hipMemsetAsync(c->d_hist, 0, WT_HIST_BINS * sizeof(uint32_t), c->stream));
        if (window) {
            // the riding histogram sits in a predicted window of 21-bit keys: its step fixes 21 bits at once
            hipLaunchKernelGGL(wt_select_window_step_kernel, dim3(1), dim3(256), 0, c->stream, c->d_hist, st, (const uint32_t *)hist_base_word(c));
            WT_HIP(hipGetLastError());
        } else {
            WT_TRY(select_pass(p, b, st, 0u, 20, 0x7ffu, 0, have_hist));
            WT_TRY(select_pass(p, b, st, 0x7ff00000u, 10, 0x3ffu, 0));
        }
        WT_TRY(select_pass(p, b, st, 0x7ffffc00u, 0, 0x3ffu, 1));
        WT_HIP(hipMemcpyAsync((char *)c->h_pinned + 64, st, sizeof(WtSelectState), hipMemcpyDeviceToHost, c->stream));
        WT_HIP(hipStreamSynchronize(c->stream));             // the one host round trip of the select
        res_st = *(const WtSelectState *)((const char *)c->h_pinned + 64);
        return 0;
    };
    // No riding histogram (a plane of a bilateral / recursive / generic transform, or an edited one):
    // the same window, placed from a 4096-sample of the plane itself, lets ONE pass bin the top 21 bits
    // - two passes over the plane instead of three.
    bool window = windowed;
    if (!pre && g_opt_hist_window && p->nranks == 1 && N >= ((int64_t)1 << 20) && p->g.H >= 64 && p->g.W >= 64) {
        WT_HIP(hipMemsetAsync(c->d_hist, 0, WT_HIST_BINS * sizeof(uint32_t), c->stream));
        uint32_t *keys = (uint32_t *)c->d_partials;          // 16 KB of the reduction scratch (stream-ordered use)
        {
            ProfScope ps(c, "wt_median_window_kernel");
            hipLaunchKernelGGL(wt_plane_sample_kernel<float>, dim3(64), dim3(64), 0, c->stream, (const float *)b, p->g.nrows, p->g.W, p->g.P, keys);
            hipLaunchKernelGGL(wt_median_window_kernel, dim3(1), dim3(1024), 0, c->stream, (const uint32_t *)keys, hist_base_word(c));
        }
        {
            ProfScope ps(c, "wt_hist_kernel");
            const int X4 = (p->g.W + 3) / 4, nchunk = (X4 + 256 * WT_HIST_UNROLL - 1) / (256 * WT_HIST_UNROLL);
            const int64_t nitems = (int64_t)p->g.nrows * nchunk;
            hipLaunchKernelGGL((wt_hist_kernel<4, true>), dim3((unsigned)std::min<int64_t>(nitems, 4 * c->num_cus)), dim3(256), 0, c->stream,
                               (const float *)b, p->g.nrows, p->g.P / 4, p->g.W, 0u, (const WtSelectState *)st, 10, 0x7ffu, c->d_hist,
                               (const uint32_t *)hist_base_word(c));
        }
        WT_HIP(hipGetLastError());
        window = true;
    }
    WT_TRY(run(pre || window, window));
    if (window && res_st.failed == 3) WT_TRY(run(false, false));      // the window missed the median: the ordinary three passes
    if (res_st.failed) WT_FAIL("wt_abs_median: rank %lld not found (NaN input?)", (long long)klo);
    const int64_t cum_le = (int64_t)res_st.cum_le;           // elements <= v_lo
    const uint32_t ulo = res_st.prefix;
    uint32_t uhi = ulo;
    if ((N & 1) == 0 && cum_le < klo + 2) {
        // the upper median is the smallest element strictly greater than v_lo
        uint32_t *res = c->d_hist + WT_HIST_BINS;
        WT_HIP(hipMemsetAsync(res, 0xff, sizeof(uint32_t), c->stream));
        {
            ProfScope ps(c, "wt_min_greater_kernel");
            hipLaunchKernelGGL(wt_min_greater_kernel, dim3(std::min(p->g.nrows, 2048)), dim3(256), 0, c->stream, b, p->g.nrows, p->g.P / 4, p->g.W, ulo, res);
        }
        WT_HIP(hipGetLastError());
        if (p->nranks > 1) WT_NCCL(g_rccl.AllReduce(res, res, 1, NCCL_UINT32, NCCL_MIN, c->comm, c->stream));
        WT_HIP(hipMemcpyAsync(c->h_pinned, res, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        WT_HIP(hipStreamSynchronize(c->stream));
        uhi = *(const uint32_t *)c->h_pinned;
        if (uhi == 0xffffffffu) WT_FAIL("wt_abs_median: upper median not found");
    }
    float lo, hi;
    memcpy(&lo, &ulo, 4);
    memcpy(&hi, &uhi, 4);
    // np.median on float32: mean of the two middle values in float32
    *median = (N & 1) ? lo : (lo + hi) / 2.0f;
    return 0;
}

#include "wt_f64.h"
