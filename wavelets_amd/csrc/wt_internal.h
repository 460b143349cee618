// Internal host-side structures of libwatroo_hip.so (gfx950 only; no compatibility layers).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <thread>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/watroo_hip.h"

// ------------------------------------------------------------------ error plumbing
void wt_set_error(const char *fmt, ...);

#define WT_FAIL(...)               \
    do {                           \
        wt_set_error(__VA_ARGS__); \
        return 1;                  \
    } while (0)

#define WT_HIP(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            wt_set_error("HIP error %d (%s) at %s:%d: %s", (int)e_, hipGetErrorString(e_), \
                         __FILE__, __LINE__, #expr);                                   \
            return 2;                                                                  \
        }                                                                              \
    } while (0)

#define WT_TRY(expr)          \
    do {                      \
        int rc_ = (expr);     \
        if (rc_) return rc_;  \
    } while (0)

// ------------------------------------------------------------------ geometry shared with kernels
// One row strip of a global H x W image resident on this GPU.  `base` pointers handed to
// kernels point at LOCAL row 0 (global row row0); rows [-halo, nrows+halo) are addressable.
struct Geo {
    int W;      // valid pixels per row
    int P;      // row pitch in floats (multiple of 4)
    int H;      // global image height
    int row0;   // global row index of local row 0
    int nrows;  // rows owned by this strip
    int halo;   // margin rows allocated above and below
    int border; // 0: symmetric reflection (cv2.BORDER_REFLECT); 1: symmetric reflection WITHIN each
                // polyphase component of the operator's dilation (atrous_recursive, wavelets.py:330-406);
                // 2: 'mirror' (no edge duplication) - the 1-D branch, wavelets.py:66-69
};

// ------------------------------------------------------------------ RCCL (dlopen'ed lazily: the
// library is 570 MB; single-GPU users never page it in)
struct RcclApi;

struct ProfEntry {
    int64_t calls = 0;
    double ms = 0.0;
};

struct PendingEvent {
    std::string name;
    hipEvent_t a, b;
};

struct wt_ctx {
    // entry points that use this context (its stream, scratch buffers, plans) take this lock:
    // calls from several host threads are serialised per context (ctypes releases the GIL)
    std::recursive_mutex mu;
    int device = 0;
    int num_cus = 256;                      // hipDeviceAttributeMultiprocessorCount (MI355X: 256)
    // warm-up of the runtime's lazy parts (copy queues, code objects) beside the caller's first steps
    // (wt_core.hip: ctx_warm); joined by the first entry point that takes the lock (WtGuard)
    std::thread *warm = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    // profiling
    bool profiling = false;
    std::vector<PendingEvent> pending;
    std::vector<hipEvent_t> event_pool;
    std::map<std::string, ProfEntry> prof;
    std::vector<std::string> prof_order;
    // comm
    int rank = 0, nranks = 1;
    void *comm = nullptr;  // ncclComm_t
    // second (high-priority) stream for halo exchanges that run beside a pass, and the events that
    // order it against `stream` (created with the communicator)
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_to_comm = nullptr, ev_from_comm = nullptr;
    // Side stream (round 5): per-scale work that only depends on planes the main stream has ALREADY produced
    // runs here, beside the kernels the main stream still has queued - the wow update of scale s and the MAD
    // estimate on w_0 beside the bilateral filter of the later scales (memory-bound beside VALU-bound).
    // side_pending: work has been queued here since the last join (every main-stream access to a plane joins
    // first: plane_base); in_side: an entry point is issuing its launches on the side stream right now
    // (`stream` IS the side stream for its duration, WtSideScope).  Created on first use.
    // INVARIANT the overlap rests on: while side_pending, the main stream touches (a) planes only through plane_base()
    // (which joins the side stream first and ends the overlap) and (b) none of the context scratch below (d_hist,
    // d_partials, h_pinned) nor the planes the side work writes (SCRATCH(3), tmp[0], PLANE_OUT) - the bilateral
    // kernels queued on the main stream read and write only the plane pointers they were launched with.  An entry
    // point that reaches plane data or the context scratch some other way (cached pointers, FFT state) must call
    // wt_side_join(ctx) first; wt_plane_sum_early additionally assumes that planes [0, count) were all updated on
    // the side stream since the transform (overlap_ok && side_pending).
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_side_done = nullptr;
    bool side_pending = false;
    int in_side = 0;
    // transfer streams of the pipelined host-to-host call (wt_decompose_sum_host), created on first use
    hipStream_t xfer_in = nullptr, xfer_out = nullptr;
    // small device/host scratch for selects & reductions
    uint32_t *d_hist = nullptr;   // 2048 bins + extras
    double *d_partials = nullptr; // reduction partials
    int partial_blocks = 0;
    void *h_pinned = nullptr;     // 64 KiB pinned host scratch
    float *d_psf = nullptr;       // PSF taps of wt_filter2d (grown on demand)
    size_t d_psf_cap = 0;         // floats
    // tap list of the generic operator (wt_taps_conv / wt64_taps_conv): 3 int32 offsets + one
    // double-sized weight slot per tap, grown on demand
    void *d_taps = nullptr;
    size_t d_taps_cap = 0;        // taps
    // Marker a fused first pass leaves when it has histogrammed the first radix level of |plane| of
    // `prehist_plan` into d_hist (flag bit4 of wt_decompose / wt_decompose_pass).  wt_abs_median of
    // that plane then skips its first pass over the plane.  Dropped by any access to the plane
    // through plane_base (conservative: reads too), by any wt_abs_median and with the plan.
    const void *prehist_plan = nullptr;          // a wt_plan or a wt_plan64
    int prehist_plane = 0;
    bool prehist_ran = false;     // set by the launch of the histogram variant (per entry point)
    bool prehist_windowed = false; // the riding histogram is the windowed form (wt_median_window_kernel placed it)
    // candidate list of the float64 median select (wt64_abs_median): 64-bit keys + a counter word
    unsigned long long *d_cand = nullptr;
    size_t d_cand_cap = 0;        // keys
    // scattered planes (plan_alloc): cleared PER CONTEXT when the virtual-memory API fails on this
    // device (plain hipMalloc from then on); the reason is kept for wt_plan_memory / wt_last_error
    bool vmm_disabled = false;
    std::string vmm_reason;
};

#define WT_MAX_CUSTOM_TAPS 15
#define WT_MAX_SUM_PLANES 16      // planes one wt_plane_sum / wt_denoise_sum launch folds

// Device state of the FFT path of one plan: twiddle tables of both lengths, two work arrays and the
// kernel spectrum (stored TRANSPOSED, W x H, the layout the forward transform ends in).
struct WtFftState {
    int H = 0, W = 0;
    void *tw_w = nullptr, *tw_h = nullptr, *a = nullptr, *b = nullptr, *spec = nullptr;
    bool have_spec = false;
};



struct wt_plan {
    wt_ctx *ctx = nullptr;
    Geo g{};
    int family = WT_B3SPLINE;
    int max_level = 0;
    int rank = 0, nranks = 1;
    size_t plane_floats = 0;                // (nrows + 2*halo) * P
    // Plane buffers (first margin row).  Each plane is its own allocation, skewed by a
    // different multiple of `skew_floats` so that the same pixel of different planes does not
    // map to the same HBM channel/bank (planes are otherwise exactly 2^k bytes apart).
    std::vector<float *> coef;              // max_level+1 planes
    float *input = nullptr, *out = nullptr;
    float *scratch[WT_NUM_SCRATCH] = {nullptr};
    std::vector<void *> raw_allocs;         // what hipFree gets
    size_t skew_floats = 0;
    int n_allocs = 0;
    int scatter = 0;                        // the "scatter" option when the plan was created: its planes keep that placement
    int scatter_strips = 0;                 // ... and the "scatter_strips" option (strip plans: mapped planes or plain hipMalloc)
    // scattered planes (hipMem* virtual memory management): every plane is a contiguous VIRTUAL range
    // mapped onto physical chunks taken from a shuffled pool (see plan_alloc)
    struct VmmPlane {                       // one hipMemMap of vmm_gran bytes per chunk
        void *va;
        size_t size;
        std::vector<hipMemGenericAllocationHandle_t> chunks;
    };
    std::vector<VmmPlane> vmm_planes;
    std::vector<hipMemGenericAllocationHandle_t> vmm_pool;   // created, not yet mapped (idle HBM: wt_plan_trim)
    size_t vmm_gran = 0;
    uint64_t vmm_seed = 0x9e3779b97f4a7c15ull;   // shuffle stream (advances: every refill deals differently)
    size_t raw_bytes = 0;                   // bytes behind raw_allocs (hipMalloc'ed planes, stage)
    float *vmm_stage = nullptr;             // hipMalloc'ed bounce plane: hipMemcpy2D does not cross mapped chunks
    void *istage = nullptr;                 // integer / byte-swapped image on its way into a plane (wt_upload_int)
    size_t istage_cap = 0;
    // user-defined scaling function (wt_plan_set_taps): odd number of 1-D taps, 0 = built-in family
    int ntaps = 0;
    float taps[WT_MAX_CUSTOM_TAPS] = {0};
    WtFftState fft;                         // circular products of richardson_lucy(fft=True), wt_fft.h (buffers in raw_allocs)
    // events recorded on the main stream behind scale s of the last wt_decompose_bilateral ("w_s is written");
    // overlap_scales of them are current as long as nothing else has touched the plan (overlap_ok)
    std::vector<hipEvent_t> scale_ev;
    int overlap_scales = 0;
    bool overlap_ok = false;
};

// float64 engine (wt_f64.h): double planes; the fused double passes (wt_fused_tu.hip) launch on it too
#define WT64_NUM_SCRATCH 32
#define WT64_MAX_TAPS 15

struct wt_plan64 {
    wt_ctx *ctx = nullptr;
    Geo g{};                                   // P = pitch in doubles (even)
    int max_level = 0;
    double taps[WT64_MAX_TAPS] = {0};
    int ntaps = 0;
    std::vector<double *> coef;
    double *input = nullptr, *out = nullptr;
    double *scratch[WT64_NUM_SCRATCH] = {nullptr};
    double *tmp[3] = {nullptr, nullptr, nullptr};   // private temporaries of the filters (no plane id)
    double *psf = nullptr;                          // PSF taps of wt64_filter2d
    size_t psf_cap = 0;
    void *istage = nullptr;                         // integer image on its way into a plane (wt64_upload_int)
    size_t istage_cap = 0;
    std::vector<void *> allocs;
    WtFftState fft;                                 // as wt_plan::fft (buffers in allocs)
    std::vector<hipEvent_t> scale_ev;               // as wt_plan::scale_ev
    int overlap_scales = 0;
    bool overlap_ok = false;
};

// ------------------------------------------------------------------ side stream (wt_core.hip)
int wt_side_join(wt_ctx *c);                        // main stream waits for everything queued on the side stream
int wt_side_begin(wt_ctx *c, hipEvent_t after);     // route the context's launches to the side stream, behind `after`
void wt_side_end(wt_ctx *c);
struct WtSideScope {                                // RAII form (on = false: nothing); ok() is false when the streams could not be set up
    wt_ctx *c;
    bool on;
    int rc;
    WtSideScope(wt_ctx *ctx, hipEvent_t after, bool enable) : c(ctx), on(enable), rc(enable ? wt_side_begin(ctx, after) : 0) {}
    ~WtSideScope() { if (on && !rc) wt_side_end(c); }
    WtSideScope(const WtSideScope &) = delete;
    WtSideScope &operator=(const WtSideScope &) = delete;
    bool ok() const { return rc == 0; }
};
bool wt_wow_overlap_enabled();
int wt_scale_events(wt_ctx *c, std::vector<hipEvent_t> &ev, int n);   // at least n events in ev

// Profiling bracket: records events around a kernel launch when ctx->profiling.
struct ProfScope {
    wt_ctx *ctx;
    const char *name;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st = nullptr;
    ProfScope(wt_ctx *c, const char *n, hipStream_t s = nullptr);
    ~ProfScope();
};
