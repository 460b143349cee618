// Host-side interface of the fused multi-scale passes (wt_fused.h): argument block, row ranges, which
// passes exist, and the entry points of the translation units the instantiations are compiled in.
// The 200-odd instantiations of wt_fused_kernel take minutes to compile in one piece; they are built
// as one group per (element type, taps, variant) in wt_fused_tu.hip - 16 translation units that
// compile side by side - and the host units see only the functions declared here.
#pragma once
#include <hip/hip_runtime.h>

#include "wt_internal.h"

#define WT_FUSED_MAX_SCALES 4      // four only for the 3-tap family (register window 2 * 15 float4)
#define WT_FUSED_MAX_FIRST_SCALE 3

template <typename T>
struct FusedArgsT {
    const T *in;                           // c_{s0}, local row 0
    T *out_c;                              // c_{s0+NS}
    T *out_w[3];                           // w_{s0+a}, a < 3 (the fourth plane of a four-scale pass: out_w3, last field)
    Geo g;
    int Vx;       // valid (stored) pixels per x-strip, multiple of 32
    int S;        // chain steps stored per chunk
    int chunks;   // chunks per chain
    // accumulate variants (ACC != 0): the plane sum np.sum(planes, axis=0) carried through the
    // passes in plane order - p_in = w_0 + ... + w_{s0-1} (nullptr for the first pass),
    // p_out = p_in + w_{s0} + ... + w_{s0+NS-1} (+ c_{s0+NS} in the last pass); may alias p_in
    const T *p_in;
    T *p_out;
    // rows stored by this launch: up to two ranges [rlo, rhi) of strip-local rows (blockIdx.z);
    // a whole pass is the single range [0, nrows).  Splitting a pass into its edge rows and its
    // interior lets the halo exchange of the NEXT pass overlap with the interior (multi-GPU).
    int rlo[2], rhi[2];
    // ACC == 3 (plain pass + first level of the exact-median select): 2048-bin histogram of the top
    // 11 magnitude bits of the first detail plane's stored pixels, added to these global bins
    uint32_t *hist;
    int debug;    // ablation switches (WT_FUSED_DEBUG): 1 = drop stores,
                  // 2 = loads re-read one row, 4 = no filtering (same loads/stores), 16 = with 4: do
                  // not issue the predicated-off stores
    T *out_w3;      // w_{s0+3} of a four-scale pass (3-tap family); LAST on purpose: the older fields keep their offsets
    // ACC == 3: nullptr = bins are the top 11 magnitude bits; else the WINDOWED form (round 4) - bin =
    // clamp((bits >> 10 [float] or 41 [double]) - *hist_base, 0, 2047): 2046 bins of 21- / 22-bit resolution
    // around a predicted median (wt_median_window_kernel), everything below / above in bins 0 / 2047
    const uint32_t *hist_base;
};
typedef FusedArgsT<float> FusedArgs;

// The fused march addresses the rows of a chunk with 31-bit byte offsets (fixed store
// descriptors): the shortest chunk of the widest-dilation pass (D = 64: ~48 steps of 64 rows)
// must stay below 2 GiB, i.e. rows up to ~174 000 pixels.  Wider images take the per-scale kernels.
static inline bool wt_fused_supported(const wt_plan *p) { return (int64_t)p->g.P * 4 * 64 * 48 < ((int64_t)1 << 31); }
static inline bool wt_fused_supported_bytes(int64_t pitch_bytes) { return pitch_bytes * 64 * 48 < ((int64_t)1 << 31); }

// Rows a launch stores (strip-local): n = 1 or 2 ranges.  reserve = compute units the chunk
// search leaves free (for the RCCL kernels of an exchange running beside the launch).
// A/B switch (wt_set_option "fused_fast"): 0 forces the generic addressing of the fused passes
extern int g_opt_fused_fast;      // defined in wt_transform.hip (env WT_FUSED_NO_FAST)

struct FusedRows {
    int n = 0;
    int lo[2] = {0, 0}, hi[2] = {0, 0};
    int reserve = 0;
    int part = 0;      // profiling label of a split pass (multi-GPU): 1 = "/interior", 2 = "/edge"
};

static inline bool wt_fused_has_pass(int s0, int ns, int family = WT_B3SPLINE)
{
    if (family == WT_TRIANGLE && ns == 4 && (s0 == 0 || s0 == 4)) return true;
    if (ns == 1 && (s0 == 3 || s0 == 6)) return true;    // the single scale that ends a 4- or 7-scale schedule
    return (s0 == 0 && (ns == 2 || ns == 3)) || (s0 == 3 && (ns == 2 || ns == 3)) || (s0 == 6 && ns == 2);
}


// One function per translation unit (wt_fused_tu.hip): the passes of one element type, tap count and
// variant (acc: 0 plain, 1 carries the plane sum, 2 last pass of a sum, 3 plain + first level of the
// median select - float only).
#define WT_FUSED_TU_DECL(K, ACC)                                                                                       \
    int wt_fused_tu_f32_k##K##_acc##ACC(wt_plan *p, const FusedArgs &a, int s0, int ns, const FusedRows &rows);
WT_FUSED_TU_DECL(5, 0) WT_FUSED_TU_DECL(5, 1) WT_FUSED_TU_DECL(5, 2) WT_FUSED_TU_DECL(5, 3)
WT_FUSED_TU_DECL(3, 0) WT_FUSED_TU_DECL(3, 1) WT_FUSED_TU_DECL(3, 2) WT_FUSED_TU_DECL(3, 3)
#undef WT_FUSED_TU_DECL
#define WT_FUSED_TU_DECL(K, ACC)                                                                                       \
    int wt_fused_tu_f64_k##K##_acc##ACC(wt_plan64 *p, const FusedArgsT<double> &a, int s0, int ns, const FusedRows &rows);
WT_FUSED_TU_DECL(5, 0) WT_FUSED_TU_DECL(5, 1) WT_FUSED_TU_DECL(5, 2) WT_FUSED_TU_DECL(5, 3)
WT_FUSED_TU_DECL(3, 0) WT_FUSED_TU_DECL(3, 1) WT_FUSED_TU_DECL(3, 2) WT_FUSED_TU_DECL(3, 3)
#undef WT_FUSED_TU_DECL

// acc: see wt_fused_dispatch_acc; p_in / p_out only for acc != 0
static int wt_fused_launch(wt_plan *p, const float *in, float *out_c, float **out_w, int s0, int ns,
                           int acc = 0, const float *p_in = nullptr, float *p_out = nullptr,
                           const FusedRows &rows = FusedRows(), uint32_t *hist = nullptr, const uint32_t *hist_base = nullptr)
{
    FusedArgs a{};
    a.in = in;
    a.out_c = out_c;
    for (int i = 0; i < ns && i < 3; ++i) a.out_w[i] = out_w[i];
    a.out_w3 = ns > 3 ? out_w[3] : nullptr;
    a.g = p->g;
    a.p_in = p_in;
    a.p_out = p_out;
    a.hist = hist;
    a.hist_base = hist_base;
    const bool b3 = p->family == WT_B3SPLINE;
    if (acc == 3) return b3 ? wt_fused_tu_f32_k5_acc3(p, a, s0, ns, rows) : wt_fused_tu_f32_k3_acc3(p, a, s0, ns, rows);
    if (acc == 1) return b3 ? wt_fused_tu_f32_k5_acc1(p, a, s0, ns, rows) : wt_fused_tu_f32_k3_acc1(p, a, s0, ns, rows);
    if (acc == 2) return b3 ? wt_fused_tu_f32_k5_acc2(p, a, s0, ns, rows) : wt_fused_tu_f32_k3_acc2(p, a, s0, ns, rows);
    return b3 ? wt_fused_tu_f32_k5_acc0(p, a, s0, ns, rows) : wt_fused_tu_f32_k3_acc0(p, a, s0, ns, rows);
}
