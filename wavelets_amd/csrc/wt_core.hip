// libwatroo_hip.so - host side of the C ABI declared in include/watroo_hip.h, unit 1 of 4: errors, profiling,
// RCCL (loaded on demand), the side stream, contexts, plans and their memory, host transfers, halo exchange.
// gfx950 only.
#include <dlfcn.h>
#include <sys/mman.h>

#include <algorithm>
#include <thread>
#include <chrono>
#include <cstdarg>
#include <cstdlib>

#include "wt_internal.h"
#include "wt_host.h"
#include "wt_kernels_common.h"
#include "wt_kernels_core.h"
#include "wt_rccl_group.h"
#include "wt_unit_probe.h"

WT_UNIT_PROBE_DEFINE

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
RcclApi g_rccl;
int rccl_load()
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
int g_opt_wow_overlap = getenv("WT_NO_WOW_OVERLAP") ? 0 : 1;   // wt_set_option("wow_overlap", 0/1)
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

// ---------------------------------------------------------------------------------------------
// Warm-up thread of a context (round 5).  tools/first_call.py: of the 40-52 ms a process's FIRST denoise(img)
// at 8192^2 took after the context existed, 22 ms were the runtime setting up its copy machinery on the first
// transfer in each direction (upload 16.5 ms against 4.7 in steady state, download 14.6 against 4.8 - the same
// with a 256 KiB image, so not a matter of size) and 15 ms loading the code objects of the five units the call
// launches from.  Neither needs the caller's data.
// wt_ctx_create starts a thread that does one small transfer in each direction the numpy-to-numpy calls take
// (pageable host memory to the device, the device to a page-locked block) through the calls the plans use, then
// loads the three host-side units (wt_unit_probe.h; ~1 ms each): 18 ms.  It runs beside whatever the caller does
// between creating the context and its first call on it (reading its image, say); the first entry point that
// takes the context's lock joins it (WtGuard), so nothing of the library ever runs beside it - two threads inside
// the runtime's lazy set-up only queue behind each other (measured: the caller's first upload 21 ms instead of
// 14).  A first call that finds the thread finished takes 14 ms instead of 33.
// WATROO_HIP_NO_WARMUP=1 switches it off; WATROO_HIP_WARMUP_TRACE=1 prints what it spent.
// What was tried and dropped: loading all 21 units up front (60 ms, and a caller that does not wait gets its
// own launches queued behind units it never uses: first call 66 ms instead of 42); a second thread per plan
// loading the fused passes of its family and the per-scale unit (a caller that syncs first: 18 ms instead of 14,
// one that does not: 40 instead of 33 - the loads contend with the caller's own upload and launches, and the
// fused units cost 0.4-0.9 ms each on demand anyway).  What did help without any thread: the per-scale float32
// kernels used to sit in the unit of the launch code (5 MB, 10.5 ms to load, paid by every first transform whether
// it used them or not) - now wt_stencil32.hip, loaded by the first per-scale launch (first call 42 -> 33 ms).
// ---------------------------------------------------------------------------------------------
#define WT_UNIT_DECL(name) int wt_unit_load_##name();
WT_UNITS(WT_UNIT_DECL)
#undef WT_UNIT_DECL
struct WtUnit {
    const char *name;
    int (*load)();
};
#define WT_UNIT_ROW(name) {#name, wt_unit_load_##name},
static const WtUnit kUnits[] = {WT_UNITS(WT_UNIT_ROW)};
#undef WT_UNIT_ROW

extern "C" int wt_unit_count(void) { return (int)(sizeof(kUnits) / sizeof(kUnits[0])); }
extern "C" const char *wt_unit_name(int i) { return i >= 0 && i < wt_unit_count() ? kUnits[i].name : nullptr; }

static bool warm_trace() { static const bool v = getenv("WATROO_HIP_WARMUP_TRACE") && atoi(getenv("WATROO_HIP_WARMUP_TRACE")); return v; }
static double warm_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void load_units(int device, std::vector<std::string> names)
{
    if (hipSetDevice(device) != hipSuccess) return;
    for (const auto &n : names)
        for (const auto &u : kUnits)
            if (n == u.name) {
                const double t0 = warm_now();
                (void)u.load();
                if (warm_trace()) fprintf(stderr, "[watroo_hip warm-up] unit %-20s %7.2f ms (at %.2f)\n", u.name, warm_now() - t0, t0);
            }
    (void)hipGetLastError();
}

static void ctx_warm(wt_ctx *c)
{
    // (the context's own stream and scratch buffers: no entry point runs on this context before the thread is
    //  joined - WtGuard; a stream and buffers of its own cost the thread another 10 ms)
    if (hipSetDevice(c->device) != hipSuccess) return;
    const size_t rows = 64, row_bytes = 1024;                 // 64 KiB: h_pinned, d_partials
    std::vector<char> pageable(rows * row_bytes, 0);
    double t0 = warm_now();
    auto lap = [&](const char *what) {
        const double t1 = warm_now();
        if (warm_trace()) fprintf(stderr, "[watroo_hip warm-up] %-28s %7.2f ms\n", what, t1 - t0);
        t0 = t1;
    };
    (void)hipMemcpy2DAsync(c->d_partials, row_bytes, pageable.data(), row_bytes, row_bytes, rows, hipMemcpyHostToDevice, c->stream);
    (void)hipStreamSynchronize(c->stream);
    lap("pageable -> device");
    (void)hipMemcpy2DAsync(c->h_pinned, row_bytes, c->d_partials, row_bytes, row_bytes, rows, hipMemcpyDeviceToHost, c->stream);
    (void)hipStreamSynchronize(c->stream);
    lap("device -> page-locked");
    (void)hipGetLastError();
    load_units(c->device, {"core", "transform", "apps"});
}

// (called with the context's lock held: WtGuard, wt_ctx_destroy)
void wt_ctx_warm_join(wt_ctx *c)
{
    if (!c || !c->warm) return;
    c->warm->join();
    delete c->warm;
    c->warm = nullptr;
}

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
    static_assert((kPartialBlocks * 4 + 8) * sizeof(double) >= 65536, "ctx_warm copies 64 KiB through d_partials");
    if (!(getenv("WATROO_HIP_NO_WARMUP") && atoi(getenv("WATROO_HIP_NO_WARMUP")))) c->warm = new std::thread(ctx_warm, c);
    *out = c;
    return 0;
}

extern "C" int wt_ctx_destroy(wt_ctx *c)
{
    if (!c) return 0;
    wt_ctx_warm_join(c);
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
// A/B switch (wt_set_option "tri4"): four-scale passes of the 3-tap family for level >= 8
int g_opt_tri4 = getenv("WT_NO_TRI4") ? 0 : 1;
// planes over shuffled physical chunks (plan_alloc): chunks are created in groups worth this many
// planes; 0 = plain hipMalloc per plane (contiguous planes: interop through wt_plane_ptr)
int g_opt_scatter = getenv("WT_SCATTER") ? atoi(getenv("WT_SCATTER")) : 4;
// strip plans (nranks > 1) too: wt_set_option("scatter_strips", 1) / WT_SCATTER_STRIPS=1; bench.py --gpus N
// measures both placements on the real transport and keeps the faster
int g_opt_scatter_strips = getenv("WT_SCATTER_STRIPS") ? atoi(getenv("WT_SCATTER_STRIPS")) : 0;

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
    void *raw = nullptr;
    // Planes whose physical memory is NOT one contiguous run (default; WT_SCATTER=0 restores plain
    // hipMalloc, WT_SCATTER=c sets the group size).  Measured on MI355X (profiles/r02_d): the same
    // binary runs the headline step in 0.61-0.66 ms when the planes' 2-MiB pages are scattered and in
    // 0.76 ms when the planes lie physically back to back (one allocation carved into planes - the round-2
    // WT_ARENA experiment, profiles/r02_d_arena_*.txt, removed in round 6 - or a freshly booted box, whose
    // allocator hands out consecutive blocks: the "slow hosts" of round 1).  The passes write the
    // same pixel of 5 planes side by side; with planes a power of two apart those addresses differ
    // only in bits the HBM channel / bank hash folds away, and the streams fight over the same banks.
    // So: physical chunks of the allocation granularity (2 MiB) are created in groups worth
    // `scatter` planes and dealt to the planes in a shuffled order (fixed seed); each plane stays one
    // contiguous VIRTUAL range (hipMemAddressReserve / hipMemMap).  Small planes (< 8 MiB) stay on
    // hipMalloc: nothing to gain, and a map call per chunk to lose.
    const int scatter = p->scatter;      // wt_set_option("scatter", n) / WT_SCATTER at the time the plan was created
    // Strip plans keep plain hipMalloc unless the "scatter_strips" option (WT_SCATTER_STRIPS=1) was on when the plan
    // was created: RCCL reads and writes the planes of a strip, and its xGMI transport has never run on mapped
    // memory here (the socket transport of the one-GPU rank test has, green).  Who decides: bench.py --gpus N times
    // both placements behind the ramp self-check; parallel.StripTransform(planes="auto") does the same once per
    // (shape, ranks) - three steps of each, results compared through the all-reduced moments - and keeps the faster.
    if (!raw && scatter > 0 && !p->ctx->vmm_disabled && need >= ((size_t)8 << 20) && (p->nranks == 1 || p->scatter_strips)) {
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
int plane_base(wt_plan *p, int id, float **base)
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

static thread_local int t_scatter_override = -1;      // wt_plan_create_placed: the placement of the plan being created

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
    p->scatter = t_scatter_override >= 0 ? t_scatter_override : g_opt_scatter;
    p->scatter_strips = g_opt_scatter_strips;
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

extern "C" int wt_plan_create_placed(wt_ctx *ctx, int64_t H, int64_t W, int family, int max_level, int scatter, wt_plan **out)
{
    WtGuard guard_(ctx_of(ctx));
    t_scatter_override = scatter < 0 ? 0 : (scatter > 16 ? 16 : scatter);
    const int rc = wt_plan_create_strip(ctx, H, W, family, max_level, 0, H, 0, 0, 1, out);
    t_scatter_override = -1;
    return rc;
}

void destroy_events(std::vector<hipEvent_t> &ev)
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
bool is_vmm(const wt_plan *p, const float *b)
{
    for (auto &v : p->vmm_planes)
        if ((const char *)b >= (const char *)v.va && (const char *)b < (const char *)v.va + v.size) return true;
    return false;
}
int vmm_stage(wt_plan *p, float **stage)
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
int vmm_copy(wt_plan *p, float *dst, const float *src)
{
    const int64_t n4 = (int64_t)p->g.nrows * p->g.P / 4;
    hipLaunchKernelGGL(wt_copy_kernel, dim3((unsigned)std::min<int64_t>((n4 + 255) / 256, 2048)), dim3(256), 0, p->ctx->stream, dst, src, n4);
    WT_HIP(hipGetLastError());
    return 0;
}
// rows x cols floats, device to device, on `st`
int copy2d(wt_plan *a, wt_plan *b, float *dst, size_t dpitch, const float *src, size_t spitch, size_t cols,
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

bool try_pin(const void *host, size_t bytes)
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
int halo_exchange_on(wt_plan *p, int plane, int64_t rows, hipStream_t st, const char *prof_name)
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

