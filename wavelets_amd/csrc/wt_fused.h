// Fused multi-scale a-trous passes: NS consecutive scales per HBM round trip (NS = 2, 3; 4 for the
// 3-tap family, whose window for four scales is as large as the 5-tap family's for three).
//
//   pass(s0, NS):  read c_{s0} once  ->  write w_{s0} .. w_{s0+NS-1} and c_{s0+NS}
//                  (watroo/wavelets.py:429-442 for NS consecutive iterations of the loop)
//
// Algorithmic traffic 4*(NS+2) B/pixel instead of 12*NS for one kernel per scale.
//
// Geometry (D = 2^s0 is the base dilation of the pass):
//  * y: the image rows split into D POLYPHASE CHAINS  y = q, q+D, q+2D, ...; on a chain the
//    scales s0+a have dilation 2^a chain steps.  A workgroup marches down one chunk of one
//    chain, one row per step, so the vertical filters are sliding windows held in REGISTERS
//    (per lane: (K-1)*(2^NS-1) float4 = 28 for B3/NS=3), never re-read from memory.
//  * x: a workgroup of NW waves covers NW*256 contiguous pixels of the row (lane = 4 adjacent
//    pixels = one 16-byte coalesced access), including the cumulative halo
//    hw*(2^NS-1)*D pixels on each side (rounded up to 32 pixels so that every wave access
//    covers whole 128-byte lines) whose results are discarded.  The horizontal filter
//    needs the vertically-filtered row of the neighbouring lanes: it is staged through a
//    per-scale LDS row (one ds_write_b128 + K-1 ds_read_b128 per lane per scale) - for D >= 4
//    the dilated taps are whole-lane offsets, for D = 1 the taps of dilation 1 and 2 are
//    recombined from the two adjacent lanes' float4.  The NS scales of a step are software-
//    pipelined (scale a works on the row scale a-1 produced one step earlier): one barrier per
//    row, LDS rows double-buffered by step parity.
//  * borders: the chain simply continues through reflected rows / columns; symmetric
//    extension commutes with the symmetric filters, so every intermediate scale is the exact
//    symmetric extension too.  In a multi-GPU strip the rows beyond the strip come from the
//    halo margins (RCCL exchange of the pass input) instead.
//  * latency of the cascade: output row of scale a lags the input row by hw*(2^(a+1)-1)
//    chain steps (+ a for the pipeline skew), so a chunk of S rows reads S + 2*hw*(2^NS-1)
//    rows (warm-up).  The host picks the chunk count that minimises (dispatch rounds) x (rows
//    per workgroup) - normally one round with every resident slot filled.
//  * stores are branch-free: one fixed raw buffer descriptor per plane; a row outside the chunk
//    or a halo lane gets an out-of-range ("parked") offset and the hardware range check drops
//    the store.
//  * the march is instruction-issue bound as much as memory bound: ~218
//    instructions per step, every scalar instruction in the step was paid for.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "wt_internal.h"
#include "wt_device.h"
#include "wt_fused_decl.h"


// (WtVec<T>, the double2 forms of the float4 helpers, wt_tap_s, wt_vzero / wt_vrev / wt_vfence: wt_device.h)

template <typename T, int K, int SHIFT_PX, int NLANES>
__device__ __forceinline__ typename WtVec<T>::V wt_hfilter_lds(const typename WtVec<T>::V *vrow, int gl, typename WtVec<T>::V own)
{
    // horizontal K-tap filter with taps SHIFT_PX pixels apart; vrow = the WG's LDS row of
    // vertically filtered values (one 16-byte group per lane), gl = this lane's index in the row.
    typedef typename WtVec<T>::V V;
    constexpr int PX = WtVec<T>::PX;
    constexpr int hw = K / 2;
    if constexpr (SHIFT_PX % PX == 0) {
        constexpr int LO = SHIFT_PX / PX;
        V acc;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            V v;
            if (j == hw) v = own;
            else {
                int idx = gl + (j - hw) * LO;
                idx = idx < 0 ? 0 : (idx > NLANES - 1 ? NLANES - 1 : idx);
                v = vrow[idx];
            }
            acc = (j == 0) ? f4_scale(wt_tap_s<K, T>(0), v) : f4_fma(wt_tap_s<K, T>(j), v, acc);
        }
        return acc;
    } else if constexpr (PX == 2) {
        // double: the only sub-group shift is 1 pixel (D = 1, first scale): both neighbours' pairs
        static_assert(SHIFT_PX == 1, "float64: sub-group shift is 1 px");
        const V L = vrow[gl > 0 ? gl - 1 : 0];
        const V R = vrow[gl < NLANES - 1 ? gl + 1 : NLANES - 1];
        const T e[6] = {L.x, L.y, own.x, own.y, R.x, R.y};
        T o[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            T acc = wt_tap_s<K, T>(0) * e[2 + k - hw];
#pragma unroll
            for (int j = 1; j < K; ++j) acc = fma(wt_tap_s<K, T>(j), e[2 + k + (j - hw)], acc);
            o[k] = acc;
        }
        return make_double2(o[0], o[1]);
    } else {
        static_assert(SHIFT_PX == 1 || SHIFT_PX == 2, "sub-float4 shifts are 1 or 2 px");
        const float4 L = vrow[gl > 0 ? gl - 1 : 0];
        const float4 R = vrow[gl < NLANES - 1 ? gl + 1 : NLANES - 1];
        if constexpr (SHIFT_PX == 1) {
            // The 1-pixel taps pair (own.y, own.z).  Without this fence the vectoriser carries
            // that odd pairing back through the vertical filter into the window and the row
            // loads (a third copy of every row: one more load per row, waited on at once, 1.5x
            // the vertical arithmetic and 8 more VGPRs); with it the pair is formed here.
            asm volatile("" : "+v"(own.x), "+v"(own.y), "+v"(own.z), "+v"(own.w));
        }
        const float e[12] = {L.x, L.y, L.z, L.w, own.x, own.y, own.z, own.w, R.x, R.y, R.z, R.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float acc = wt_tap<K>(0) * e[4 + k - hw * SHIFT_PX];
#pragma unroll
            for (int j = 1; j < K; ++j) acc = fmaf(wt_tap<K>(j), e[4 + k + (j - hw) * SHIFT_PX], acc);
            o[k] = acc;
        }
        return make_float4(o[0], o[1], o[2], o[3]);
    }
}

// Experiment (-DWT_FUSED_WAVEAUTO, D = 1 passes): WAVE-AUTONOMOUS horizontal taps.  Every wave
// covers its own 256 pixels including 4 halo lanes per side (the cumulative x halo of three scales
// at D = 1 is 14 px), neighbouring lanes' values come through DPP wave shifts, there is no LDS row
// and no barrier.  Same taps in the same order as wt_hfilter_lds: identical bits on the lanes
// that store.  DESIGN.md 3.1 has the measurement.
#ifdef WT_FUSED_WAVEAUTO
#define WT_FUSED_WA 1
#else
#define WT_FUSED_WA 0
#endif
__device__ __forceinline__ float wt_dpp_from_prev(float v)   // lane i <- lane i-1 (wave_shr:1)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float wt_dpp_from_next(float v)   // lane i <- lane i+1 (wave_shl:1)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ float4 wt_dpp4_prev(float4 v)
{
    return make_float4(wt_dpp_from_prev(v.x), wt_dpp_from_prev(v.y), wt_dpp_from_prev(v.z), wt_dpp_from_prev(v.w));
}
__device__ __forceinline__ float4 wt_dpp4_next(float4 v)
{
    return make_float4(wt_dpp_from_next(v.x), wt_dpp_from_next(v.y), wt_dpp_from_next(v.z), wt_dpp_from_next(v.w));
}
template <int K, int SHIFT_PX>
__device__ __forceinline__ float4 wt_hfilter_dpp(float4 own)
{
    constexpr int hw = K / 2;
    static_assert(SHIFT_PX == 1 || SHIFT_PX == 2 || SHIFT_PX == 4, "D = 1 passes only");
    const float4 L = wt_dpp4_prev(own), R = wt_dpp4_next(own);
    if constexpr (SHIFT_PX == 4) {
        float4 nb[5] = {own, own, own, own, own};       // lanes -2 .. +2
        nb[1] = L; nb[3] = R;
        if constexpr (hw == 2) { nb[0] = wt_dpp4_prev(L); nb[4] = wt_dpp4_next(R); }
        float4 acc;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float4 v = nb[2 + j - hw];
            acc = (j == 0) ? f4_scale(wt_tap<K>(0), v) : f4_fma(wt_tap<K>(j), v, acc);
        }
        return acc;
    } else {
        const float e[12] = {L.x, L.y, L.z, L.w, own.x, own.y, own.z, own.w, R.x, R.y, R.z, R.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float acc = wt_tap<K>(0) * e[4 + k - hw * SHIFT_PX];
#pragma unroll
            for (int j = 1; j < K; ++j) acc = fmaf(wt_tap<K>(j), e[4 + k + (j - hw) * SHIFT_PX], acc);
            o[k] = acc;
        }
        return make_float4(o[0], o[1], o[2], o[3]);
    }
}

// vertical window of scale A: 2^A interleaved sub-chains, K-1 stored rows each
template <typename V, int K, int A>
struct VWin {
    V w[1 << A][K - 1];
};

typedef unsigned int wt_v4u __attribute__((ext_vector_type(4)));
typedef float wt_v4f __attribute__((ext_vector_type(4)));

// Branch-free predicated 16-byte stores through raw buffer descriptors: a row or lane that must
// not be written gets an out-of-range offset and the hardware range check drops the store.
// Control flow stays uniform, so the compiler's vmcnt bookkeeping is exact (loads stay in flight
// across the stores and barriers of several steps).
// AUX = cache-policy bits of the store (gfx950: 1 = sc0, 2 = nt, 16 = sc1).
//
// Store through a descriptor that stays FIXED for the whole march (base = the row the plane
// stores at step 0, length = the chunk's byte span): the row is selected by the byte offset
// k * step_bytes folded into voff (one v_add per step shared by all planes), a row or lane that
// must not be written gets the parked offset 2^31 >= length.  No per-store scalar work besides
// the row predicate: the march is instruction-issue bound.
#define WT_FUSED_PARKED 0x80000000u
template <int AUX>
__device__ __forceinline__ void wt_bstore4v(__amdgpu_buffer_rsrc_t r, unsigned voff, float4 v)
{
    wt_v4f t = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wt_v4u, t), r, voff, 0, AUX);
}
typedef double wt_v2d __attribute__((ext_vector_type(2)));
template <int AUX>
__device__ __forceinline__ void wt_bstore4v(__amdgpu_buffer_rsrc_t r, unsigned voff, double2 v)
{
    wt_v2d t = {v.x, v.y};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wt_v4u, t), r, voff, 0, AUX);
}
__device__ __forceinline__ float4 wt_from_v4u(wt_v4u t, float4)
{
    const wt_v4f f = __builtin_bit_cast(wt_v4f, t);
    return make_float4(f.x, f.y, f.z, f.w);
}
__device__ __forceinline__ double2 wt_from_v4u(wt_v4u t, double2)
{
    const wt_v2d f = __builtin_bit_cast(wt_v2d, t);
    return make_double2(f.x, f.y);
}

// Ablation switches (FusedArgs::debug, env WT_FUSED_DEBUG) are compiled in only with
// -DWT_FUSED_ABLATION: their loop-invariant branches cost issue slots in every step.
#ifdef WT_FUSED_ABLATION
#define WT_FUSED_DBG(a) ((a).debug)
#else
#define WT_FUSED_DBG(a) 0
#endif

// The detail planes are write-once streams that nothing re-reads before the pass is over:
// nontemporal stores.  The smooth plane is the NEXT pass's input and keeps the default policy
// (up to 4096^2 it is still in the Infinity Cache when the next pass starts: 0.125 -> 0.097 ms
// for the D = 8 pass).  Measured on several MI355X hosts at 8192^2: +-1.5 % on most, but 25 %
// faster on a host where plain stores ran the passes at 0.46 ms instead of 0.33 ms, i.e. the
// streaming stores also remove most of the host-to-host spread.
#ifndef WT_FUSED_W_AUX
#define WT_FUSED_W_AUX 2
#endif
#ifndef WT_FUSED_P_AUX
#define WT_FUSED_P_AUX 0   // running sum between two passes (re-read by the next pass)
#endif
#ifndef WT_FUSED_C_AUX
#define WT_FUSED_C_AUX 0
#endif
#ifndef WT_FUSED_R_AUX
#define WT_FUSED_R_AUX WT_FUSED_W_AUX   // the finished reconstruction (last pass of a carried sum)
#endif

// Vertical half of one scale of one step: push `cur` (a row of c_{s0+A}) into the window and
// return the vertically filtered row (centred hw*2^A steps back); `cen` = matching row of
// c_{s0+A} (for the detail plane).
template <typename T, int K, int A>
__device__ __forceinline__ typename WtVec<T>::V wt_fused_vstage(VWin<typename WtVec<T>::V, K, A> &win, const int kk,
                                                                const typename WtVec<T>::V cur, typename WtVec<T>::V &cen)
{
    typedef typename WtVec<T>::V V;
    constexpr int hw = K / 2;
    constexpr int KM = K - 1;
    const int rho = kk % (1 << A);
    const int p = (kk >> A) % KM;
    V *w = win.w[rho];
    V v = f4_scale(wt_tap_s<K, T>(0), w[p]);
#pragma unroll
    for (int j = 1; j < KM; ++j) v = f4_fma(wt_tap_s<K, T>(j), w[(p + j) % KM], v);
    v = f4_fma(wt_tap_s<K, T>(KM), cur, v);
    cen = w[(p + hw) % KM];
    w[p] = cur;
    return v;
}

// The NS scales are SOFTWARE-PIPELINED across steps: in step j scale a works on the row scale
// a-1 produced in step j-1, so the NS vertical filters, the NS LDS row writes, ONE barrier and
// the NS horizontal filters of a step are mutually independent (one s_barrier per row instead
// of NS, 4*NS LDS reads in flight together).  LDS rows are double-buffered by step parity.
template <typename T, int K, int NS, int D, int NW, int PDREQ, int ACC, bool FAST>
#ifndef WT_FUSED_WPS4_K3
#define WT_FUSED_WPS4_K3 3   // 3-tap family: 148 VGPRs, three 4-wave workgroups per CU (0.345 -> 0.32 ms)
#endif
#ifndef WT_FUSED_WPS4
#define WT_FUSED_WPS4 2   // waves per SIMD requested for the 4-wave workgroup variant
#endif
// workgroups of 4 waves per CU (= waves per SIMD): 3 for the plain 3-tap passes of up to three scales,
// 1 for the four-scale accumulate variants (98 KB of LDS), else 2
#define WT_FUSED_WG4_PER_CU(K, NS, ACC) \
    ((K) == 3 && (NS) < 4 && ((ACC) == 0 || (ACC) == 3) ? WT_FUSED_WPS4_K3 : ((NS) == 4 && ((ACC) == 1 || (ACC) == 2) ? 1 : WT_FUSED_WPS4))
__global__ __launch_bounds__(NW * 64, (NW == 4 ? WT_FUSED_WG4_PER_CU(K, NS, ACC) : 2)) void wt_fused_kernel(FusedArgsT<T> a)
{
    typedef typename WtVec<T>::V V;                      // a lane's 16 bytes: float4 or double2
    constexpr int PX = WtVec<T>::PX;                     // pixels per lane
    constexpr int ALIGN_PX = 128 / (int)sizeof(T);       // pixels per 128-byte line
    // ACC: 0 = plain pass; 1 / 2 = the pass carries the plane sum (2: last pass, adds the smooth
    // plane); 3 = plain pass that also histograms |w_{s0}| (first level of wt_abs_median's select:
    // Coefficients.get_noise reads plane 0 once less)
    constexpr bool SUM = ACC == 1 || ACC == 2;
    constexpr bool HIST = ACC == 3;
    static_assert(!HIST || D == 1, "the histogram variant exists for the first pass only");
    static_assert(NS <= 3 || K == 3, "four scales per pass: 3-tap family only");
    constexpr int hw = K / 2;
    constexpr int KM = K - 1;
    constexpr int LAT_IN = hw * ((1 << NS) - 1);         // rows of input beyond a stored row
    constexpr int LAT = LAT_IN + (NS - 1);               // + pipeline skew between the scales
    // x halo rounded up to 32 pixels (128 B): with strip starts that are multiples of 32 pixels
    // every wave's 1-KiB row access is cache-line aligned (8 lines, not 9 with two half lines)
    constexpr int HX = (hw * ((1 << NS) - 1) * D + ALIGN_PX - 1) / ALIGN_PX * ALIGN_PX;
    constexpr int U = KM << (NS - 1);                    // register-rotation period
    constexpr int PD = (U % PDREQ == 0) ? PDREQ : 4;     // rows prefetched ahead
    constexpr int NL = NW * 64;                          // lanes (16-byte columns) per WG
    static_assert(U % PD == 0 && U % 2 == 0, "prefetch depth / LDS parity must divide the unroll period");

    __shared__ V vbuf[2][NS][NL];
    // output row of scale a at step t: t - a - hw*(2^(a+1)-1)
    constexpr int LAG0 = hw, LAG1 = 1 + 3 * hw, LAG2 = 2 + 7 * hw, LAG3 = 3 + 15 * hw;
    constexpr int LAGC = NS == 1 ? LAG0 : (NS == 2 ? LAG1 : (NS == 3 ? LAG2 : LAG3));
    // peeled prologue steps (see `step` below): the cascade's fill time rounded up to whole unrolled
    // bodies, at most two of them (code size); none for the single-scale passes
    // Built where the extra code does not cost registers the kernel does not have: the float D = 1
    // passes of up to three scales (B3 d1x3: 220 -> 240 VGPRs, no scratch).  The D = 8 three-scale
    // passes sit at 256 VGPRs already and the four-scale / double variants spill with it (16 - 220
    // spilled registers; still 21 with a single peeled trip, and gating only the horizontal filters
    // and stores changes nothing), so they keep the plain march.
#ifdef WT_FUSED_NO_PROLOGUE
    constexpr int PRO = 0;
#else
    constexpr bool PRO_FITS = D == 1 && NS >= 2 && NS <= 3 && sizeof(T) == 4;
    constexpr int PRO = !PRO_FITS ? 0 : (((LAT_IN + LAGC + U - 1) / U) < 2 ? ((LAT_IN + LAGC + U - 1) / U) : 2) * U;
#endif
    // ACC: the running sum of a row waits in per-lane LDS rings until the next scale's detail
    // row of the SAME image row comes out of the cascade (G1, then G2 steps later); only lanes
    // that own stored pixels take part (NV of them), nothing crosses lanes: no barrier.
    constexpr int G1 = NS > 1 ? LAG1 - LAG0 : 0, G2 = NS > 2 ? LAG2 - LAG1 : 0, G3 = NS > 3 ? LAG3 - LAG2 : 0;
    constexpr int NV = NL - 2 * HX / PX;
    __shared__ V ring[SUM && NS > 1 ? (G1 + G2 + G3) * (NV + 1) : 1];   // + one spare slot per row for the halo lanes

    __shared__ uint32_t lh[HIST ? WT_HIST_BINS : 1];
    // windowed bins: shift 10 (float) / 41 (double) and the window's first key; plain: shift 20 / 52, base 0 - one code path
    int hist_shift = 20, hist_lo = 0;
    if constexpr (HIST) {                                // (before the early exits: all waves pass the barrier)
        for (int i = threadIdx.x; i < WT_HIST_BINS; i += NL) lh[i] = 0;
        if (a.hist_base) {
            hist_shift = PX == 4 ? 10 : 41;
            hist_lo = (int)__builtin_amdgcn_readfirstlane(*a.hist_base);
        } else if (PX == 2) {
            hist_shift = 52;                             // double, plain: the exponent field
        }
        __syncthreads();
    }
    // Keys below / above the window (bins 0 and WT_HIST_BINS - 1 of the clamped index) are ~90 % of a windowed
    // histogram's samples - the window spans +-12 % around the predicted median - and as LDS atomics they all hit the
    // same two words: 64 lanes serialise on one address.  They are counted in two registers per lane instead and
    // added to their bins once, at the end of the chunk (round 6); only the in-window keys take the atomic.
    int hist_out = 0, hist_below = 0;                    // samples outside the window / below it (this lane)
    auto hist_count = [&](int rel) {                     // rel = key - first key of the window
        if ((unsigned)(rel - 1) < (unsigned)(WT_HIST_BINS - 2)) atomicAdd(&lh[rel], 1u);
        else ++hist_out;
        hist_below += rel <= 0;
    };

    const Geo g = a.g;
    const int gl = threadIdx.x;                          // lane index within the WG row
    const int X0 = blockIdx.x * a.Vx;                    // first valid pixel of this x-strip
    constexpr bool WA = WT_FUSED_WA && D == 1 && NS <= 3 && PX == 4;   // wave-autonomous horizontal taps (experiment)
    const int wa_ln = gl & 63, wa_vw = a.Vx / NW;        // lane in the wave; stored pixels per wave
    const int x = WA ? X0 + (gl >> 6) * wa_vw - 16 + 4 * wa_ln
                     : X0 - HX + PX * gl;                // this lane's first pixel (may be < 0)
    const int item = blockIdx.y;
    const int q = item % D;                              // chain phase
    const int chunk = item / D;
    if (q >= g.nrows) return;                            // whole WG exits together
    // chain elements r of this phase with rlo <= q + D*r < rhi
    const int lo = a.rlo[blockIdx.z], hi = a.rhi[blockIdx.z];
    const int ra = lo > q ? (lo - q + D - 1) / D : 0;
    const int rb = hi > q ? (hi - q + D - 1) / D : 0;
    const int r0 = ra + chunk * a.S;
    const int r1 = min(r0 + a.S, rb);
    if (r0 >= r1) return;

    // lanes that own stored pixels; a float4 that straddles W writes into the row's pitch
    // padding (allocated, never read as image data)
    const bool lane_store = WA ? (wa_ln >= 4 && 4 * (wa_ln - 4) < wa_vw && x < g.W)
                               : (x >= X0) && (x < X0 + a.Vx) && (x < g.W);
    const unsigned voff = lane_store ? (unsigned)x * (unsigned)sizeof(T) : WT_FUSED_PARKED;
    const int row_bytes = g.P * (int)sizeof(T);
    // Every lane issues ONE aligned in-bounds dwordx4 per row.
    // FAST (host: W % 4 == 0, W >= HX, H >= D * (LAT_IN + 1) - every image the benchmarks name):
    //   the reflection of an aligned 4-pixel group that lies outside the image is an aligned group
    //   read backwards, so a border lane loads that group like any other lane and reverses the
    //   four values when the row is CONSUMED (wave-uniform branch around four v_cndmask: no load
    //   sits behind a branch, every workgroup runs the same instruction stream), and a row index
    //   reflects at most once (two s_max / s_min instead of a modulo behind a branch).
    // generic: lanes whose 4 pixels are not all inside the image (reflected halo at the image
    //   border, ragged right edge) patch the value with a reflected gather under a wave-uniform
    //   branch (border waves only); rows reflect any number of times.
    const bool lane_interior = (x >= 0) && (x + PX - 1 < g.W);
    const bool wave_has_edge = !__all(lane_interior);
    // FAST with W % PX != 0 (round 6): ONE group per row straddles the right border.  Its reflected pixels lie inside
    // the same four pixels - (g0, g1, g1, g0) for W % 4 == 2, (g0, g1, g2, g2) for 3, and for 1 the four pixels that
    // END at the border read as (g3, g3, g2, g1); doubles: (g0, g0) - so it is one load and a swizzle like the
    // reversed groups, which then start at 2W - PX - x: 8-byte aligned for odd W (a 16-byte load needs 4).
    const int wrem = g.W % PX;                           // (wave-uniform)
    const bool lane_str = FAST && x < g.W && x + PX > g.W;
    const bool lane_rev = FAST && !lane_interior && !lane_str;
    const int xg = x < 0 ? -PX - x : (x >= g.W ? 2 * g.W - PX - x : (lane_str && PX == 4 && wrem == 1 ? g.W - PX : x));   // FAST: the group this lane loads
    const int xc = FAST ? min(max(xg, 0), g.P - PX) : min(max(x, 0), g.P - PX);
    const int xi0 = wt_refl(x, g.W), xi1 = wt_refl(x + 1, g.W), xi2 = wt_refl(x + (PX > 2 ? 2 : 0), g.W),
              xi3 = wt_refl(x + (PX > 2 ? 3 : 0), g.W);
    const int gy0 = g.row0 + q;                          // global row of chain element 0

    const int dbg = WT_FUSED_DBG(a);
    const int t_last = r1 - 1 + LAT_IN;                  // last input row any stored output needs
    const unsigned xoff = (unsigned)xc * (unsigned)sizeof(T);   // byte offset of this lane's aligned load
    const int H2m1 = 2 * g.H - 1;
    auto load_row = [&](int t) -> V {
        // steps past t_last only flush the pipeline / unroll padding: keep the address in range.
        // Uniform row pointer + 32-bit lane offset: one global_load_dwordx4 with an SGPR base.
        if constexpr (FAST) {
            const int gy = gy0 + D * ((dbg & 2) ? r0 : min(t, t_last));
            const int up = max(gy, ~gy);                 // -1 - gy above the image
            const int ry = min(up, H2m1 - up);           // 2H - 1 - gy below it
            const T *row = a.in + (int64_t)(ry - g.row0) * g.P;
            return *reinterpret_cast<const V *>(reinterpret_cast<const char *>(row) + xoff);
        } else {
            const T *row = a.in + (int64_t)(wt_refl(gy0 + D * ((dbg & 2) ? r0 : min(t, t_last)), g.H) - g.row0) * g.P;   // (= wt_row)
            V v = *reinterpret_cast<const V *>(reinterpret_cast<const char *>(row) + xoff);
            if (wave_has_edge) {
                if constexpr (PX == 4) {
                    if (!lane_interior) v = make_float4(row[xi0], row[xi1], row[xi2], row[xi3]);
                } else {
                    if (!lane_interior) v = make_double2(row[xi0], row[xi1]);
                }
            }
            return v;
        }
    };
    // Output rows advance by one chain step (D image rows) per iteration.  Every plane has ONE
    // descriptor for the whole march, based at the row it stores at step 0 (row t0 - LAG of the
    // chain; before the chunk, never written) and as long as the chunk's byte span (< 2 GiB,
    // host); step k adds k * step_bytes to the lane offset.  A row outside [r0, r1) parks the
    // offset instead of branching, so control flow stays uniform.
    const unsigned span = (unsigned)(r1 - r0);
    const unsigned step_bytes = (unsigned)D * (unsigned)row_bytes;
    auto row_addr0 = [&](T *base, int ro) -> uint64_t {
        return (uint64_t)base + (uint64_t)((int64_t)(q + (int64_t)D * ro) * (int64_t)row_bytes);
    };

    constexpr int A1 = NS > 1 ? 1 : 0, A2 = NS > 2 ? 2 : 0, A3 = NS > 3 ? 3 : 0;
    VWin<V, K, 0> w0;
    VWin<V, K, A1> w1;
    VWin<V, K, A2> w2;
    VWin<V, K, A3> w3;
    const V zero = wt_vzero<V>();
#pragma unroll
    for (int j = 0; j < KM; ++j) {
        w0.w[0][j] = zero;
#pragma unroll
        for (int r = 0; r < (1 << A1); ++r) w1.w[r][j] = zero;
#pragma unroll
        for (int r = 0; r < (1 << A2); ++r) w2.w[r][j] = zero;
        if constexpr (NS > 3) {
#pragma unroll
            for (int r = 0; r < (1 << A3); ++r) w3.w[r][j] = zero;
        }
    }

    const int t0 = r0 - LAT_IN;                          // first input chain index
    const int nsteps = ((r1 - r0) + LAT + LAT_IN + U - 1) / U * U;
    V pf[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) pf[i] = load_row(t0 + i);
    V c1 = zero, c2 = zero, c3 = zero;                   // rows handed from scale a to a+1
    // Stores: every plane has ONE descriptor for the whole march - base = row r0 of the chain (the
    // chunk's first stored row), length = the chunk's byte span.  At step k a plane stores row
    // k - (LAT_IN + LAG) of the chunk, i.e. the lane offset is x*4 + (k - LAT_IN - LAG) * step_bytes:
    // before the chunk that is negative (wraps to ~4 GiB), after it >= the length - the hardware
    // range check IS the row predicate, no scalar work per store.  Halo lanes park at 2^31 (the host
    // keeps a chunk's span incl. warm-up below 2^31, so a parked lane never wraps into range).
    const unsigned chunk_len = (span - 1u) * step_bytes + (unsigned)row_bytes;
    auto plane_rsrc = [&](T *base, int) -> __amdgpu_buffer_rsrc_t {
        // Length and flag words pass through an empty asm so that every descriptor owns its four
        // SGPRs: shared words would be copied into place before every store (2 s_mov each).
        unsigned len = chunk_len, flags = 0x00020000;
        asm volatile("" : "+s"(len), "+s"(flags));
        return __builtin_amdgcn_make_buffer_rsrc((void *)row_addr0(base, r0), 0, len, flags);
    };
    // byte offset of a plane's row at step 0 (negative, as unsigned): -(LAT_IN + LAG) * step_bytes
    auto lag_off = [&](int lag) -> unsigned { return 0u - (unsigned)(LAT_IN + lag) * step_bytes; };
    const unsigned o0 = lag_off(LAG0), o1 = lag_off(LAG1), o2 = lag_off(LAG2), o3 = NS > 3 ? lag_off(LAG3) : 0u, oc = lag_off(LAGC);
    const __amdgpu_buffer_rsrc_t rw0 = plane_rsrc(a.out_w[0], LAG0);
    const __amdgpu_buffer_rsrc_t rw1 = plane_rsrc(a.out_w[A1], LAG1);
    const __amdgpu_buffer_rsrc_t rw2 = plane_rsrc(a.out_w[A2], LAG2);
    // (only a four-scale pass builds the fourth descriptor: plane_rsrc pins four SGPRs)
    const __amdgpu_buffer_rsrc_t rw3 = NS > 3 ? plane_rsrc(a.out_w3, LAG3) : rw2;
    const __amdgpu_buffer_rsrc_t rc = plane_rsrc(a.out_c, LAGC);
    // ---- ACC state.  The first pass of a sum (D = 1, s0 = 0) has no incoming partial sum, every
    // later pass has one: decided at compile time (the host checks first == (s0 == 0)).
    constexpr bool PIN = SUM && D != 1;
    const __amdgpu_buffer_rsrc_t rp = plane_rsrc(SUM ? a.p_out : a.out_c, LAGC);
    unsigned koff = 0;                                   // k * step_bytes
    // incoming partial sum: the same fixed-descriptor addressing as the stores (row r0 of the chain,
    // the chunk's byte span) - a row before or after the chunk reads as 0 without touching memory
    // (its sum is never stored), so there is no clamp and no scalar address arithmetic per step
    const __amdgpu_buffer_rsrc_t rpin = plane_rsrc(PIN ? const_cast<T *>(a.p_in) : a.out_c, 0);
    // (halo lanes are parked like their stores: they read nothing)
    auto load_acc = [&](int k_ahead) -> V {              // p_in row of chain element t0 + k - LAG0, k = current step + k_ahead
        const wt_v4u t = __builtin_amdgcn_raw_buffer_load_b128(rpin, voff + koff + o0 + (unsigned)k_ahead * step_bytes, 0, 0);
        return wt_from_v4u(t, V());
    };
    V pa[PIN ? PD : 1];
    if constexpr (PIN) {
        // (with a prologue the first p_in row that matters is loaded by its step LAT_IN + LAG0 - PD)
#pragma unroll
        for (int i = 0; i < PD; ++i) pa[i] = PRO > 0 ? zero : load_acc(i);   // koff = 0 here
    }
    const int li = lane_store ? (x - X0) / PX : NV;      // slot in the ring rows; NV = the spare slot
    int i1 = 0, i2 = 0, i3 = 0;                          // ring positions (wave-uniform)

    // One chain step.
    // PROLOGUE (round 3).  The cascade fills over the first LAT_IN + LAG_last steps of a chunk: scale
    // a's horizontal filter produces a row that some stored row depends on only from step H_a on,
    // its vertical window needs real input only from step V_a on, and plane a stores from step
    // ST_a on:
    //     R_a  = hw * (2^NS - 2^(a+1))      rows of c_{a+1} beyond the chunk that later scales reach
    //     H_a  = LAG_a + LAT_IN - R_a       B3, NS = 3:  4, 13, 30
    //     V_a  = H_a - 2 * hw * 2^a                      0,  5, 14
    //     ST_a = LAT_IN + LAG_a                          16, 21, 30
    // The steady-state step does all of it at every step (3 * 30 scale-steps where 47 + 43 are
    // needed), which nobody notices while the pass waits for memory, but a grid of short chunks
    // (4096^2: 40 stored rows per chunk behind 30 warm-up steps) is bound by instruction issue.
    // The first PRO steps therefore run as peeled copies of the step in which everything that is
    // not needed yet is compiled out (k is a constant there): no vertical / horizontal filter, no
    // LDS row, no parked store, no ring traffic, no p_in load before its time.  The last PD peeled
    // steps keep the full store pattern, so that the loop is entered with the steady state's
    // vector-memory queue (what the parked stores of the vmcnt padding provided before).
    // Bit-identical: every value a stored row depends on is computed by the same instructions.
    auto step = [&](const int kb, const int kk, auto pro_tag) {
        constexpr bool PROL = decltype(pro_tag)::value;       // a peeled prologue step (kb + kk is a constant)
        const int k = kb + kk;
        const int t = t0 + k;
        constexpr int R0 = hw * ((1 << NS) - 2), R1 = hw * ((1 << NS) - 4), R2 = hw * ((1 << NS) - 8);
        constexpr int H0 = LAG0 + LAT_IN - R0, H1 = LAG1 + LAT_IN - R1, H2 = LAG2 + LAT_IN - R2, H3 = LAG3 + LAT_IN;
        constexpr int V1 = H1 - 4 * hw, V2 = H2 - 8 * hw, V3 = H3 - 16 * hw;
        constexpr int ST0 = LAT_IN + LAG0, ST1 = LAT_IN + LAG1, ST2 = LAT_IN + LAG2, ST3 = LAT_IN + LAG3;
        const bool full = !PROL || k >= PRO - PD;             // steady-state store pattern
        const bool eh0 = !PROL || k >= H0, eh1 = !PROL || k >= H1, eh2 = !PROL || k >= H2, eh3 = !PROL || k >= H3;
        const bool ev1 = !PROL || k >= V1, ev2 = !PROL || k >= V2, ev3 = !PROL || k >= V3;
        const bool es0 = full || k >= ST0, es1 = full || k >= ST1, es2 = full || k >= ST2, es3 = full || k >= ST3;
        const bool esc = full || k >= LAT_IN + LAGC;          // smooth plane / carried sum
        // step k stores row t0 + k - LAG of a plane: inside the chunk iff k - (LAT_IN + LAG) < span
        const unsigned vk = voff + koff + ((dbg & 1) ? WT_FUSED_PARKED : 0u);
        auto at = [&](int lag) -> unsigned {             // lane offset of this step's row of a plane
            if constexpr (NS > 3) return vk + (lag == LAG0 ? o0 : lag == LAG1 ? o1 : lag == LAG2 ? o2 : o3);
            else return vk + (lag == LAG0 ? o0 : lag == LAG1 ? o1 : lag == LAG2 ? o2 : oc);
        };
        V cur = pf[kk % PD];
        pf[kk % PD] = load_row(t + PD);
        if constexpr (FAST) {
            // The fence pins the reversal to THIS step: without it the (w, z) swap is scheduled
            // right behind the load it reads (same basic block), and the wave waits for every row
            // in the step that issued it - no prefetch left.
            wt_vfence(cur);
            if (wave_has_edge) {
                if (lane_rev) cur = wt_vrev(cur);
                if (wrem != 0) {
                    if (lane_str) cur = wt_vstraddle(cur, wrem);
                }
            }
        }
        V (*buf)[NL] = vbuf[kk & 1];
#ifdef WT_FUSED_ABLATION
        if (dbg & 4) {   // ablation: same loads / stores / addresses, no filtering at all
            wt_bstore4v<WT_FUSED_W_AUX>(rw0, at(LAG0), cur);
            if constexpr (NS > 1) wt_bstore4v<WT_FUSED_W_AUX>(rw1, at(LAG1), cur);
            if constexpr (NS > 2) wt_bstore4v<WT_FUSED_W_AUX>(rw2, at(LAG2), cur);
            if constexpr (NS > 3) wt_bstore4v<WT_FUSED_W_AUX>(rw3, at(LAG3), cur);
            wt_bstore4v<WT_FUSED_C_AUX>(rc, at(LAGC), cur);
            if constexpr (SUM) {
                V pv = cur;
                if constexpr (PIN) {
                    pv = pa[kk % PD];
                    pa[kk % PD] = load_acc(PD);
                }
                wt_bstore4v<(ACC == 2 ? WT_FUSED_R_AUX : WT_FUSED_P_AUX)>(rp, at(LAGC), pv);
            }
            koff += step_bytes;
            return;
        }
#endif
        V pin_cur = zero;
        if constexpr (PIN) {
            pin_cur = pa[kk % PD];
            if (!PROL || k + PD >= ST0) pa[kk % PD] = load_acc(PD);   // (rows before the chunk read as 0 anyway)
        }
        // ACC: the ring slots that come due in this step were written G1 / G2 steps ago - read
        // them before the barrier so the LDS latency hides behind the vertical filters.  Lanes
        // without stored pixels share the spare slot NV of each ring row (their sums are never
        // stored), which keeps the ring traffic free of exec-mask branches.
        V old1 = zero, old2 = zero, old3 = zero;
        if constexpr (SUM && NS > 1) {
            if (es1) old1 = ring[i1 * (NV + 1) + li];
            if constexpr (NS > 2) {
                if (es2) old2 = ring[(G1 + i2) * (NV + 1) + li];
            }
            if constexpr (NS > 3) {
                if (es3) old3 = ring[(G1 + G2 + i3) * (NV + 1) + li];
            }
        }
        V cen0, cen1, cen2, cen3, v0, v1, v2, v3;
        v0 = wt_fused_vstage<T, K, 0>(w0, kk, cur, cen0);
        if constexpr (!WA) {
            if (eh0) buf[0][gl] = v0;
        }
        if constexpr (NS > 1) {
            v1 = zero;
            if (ev1) v1 = wt_fused_vstage<T, K, A1>(w1, kk, c1, cen1);
            if constexpr (!WA) {
                if (eh1) buf[A1][gl] = v1;
            }
        }
        if constexpr (NS > 2) {
            v2 = zero;
            if (ev2) v2 = wt_fused_vstage<T, K, A2>(w2, kk, c2, cen2);
            if constexpr (!WA) {
                if (eh2) buf[A2][gl] = v2;
            }
        }
        if constexpr (NS > 3) {
            v3 = zero;
            if (ev3) v3 = wt_fused_vstage<T, K, A3>(w3, kk, c3, cen3);
            if (eh3) buf[A3][gl] = v3;
        }
        if constexpr (!WA) {
            if (eh0) __syncthreads();                         // (no scale reads the LDS rows before step H0)
        }
        V n0 = zero, d0 = zero;
        if (eh0) {
            if constexpr (WA) n0 = wt_hfilter_dpp<K, (D <= 4 ? D : 4)>(v0);
            else n0 = wt_hfilter_lds<T, K, D, NL>(buf[0], gl, v0);
            d0 = f4_sub(cen0, n0);
        }
        if (es0) wt_bstore4v<WT_FUSED_W_AUX>(rw0, at(LAG0), d0);
        if constexpr (HIST) {
            // the same predicate as the store of this row: chunk row k - (LAT_IN + LAG0) in [0, span)
            // (wave-uniform) and a lane that owns stored pixels
            if ((!PROL || k >= ST0) && (unsigned)(k - (LAT_IN + LAG0)) < span && lane_store) {
                if constexpr (PX == 4) {
                    const uint32_t b[4] = {__float_as_uint(d0.x), __float_as_uint(d0.y), __float_as_uint(d0.z),
                                           __float_as_uint(d0.w)};
#pragma unroll
                    for (int j = 0; j < 4; ++j)       // (one v_bfe_u32 for the magnitude's top bits: the march is issue-bound)
                        if (FAST || x + j < g.W) hist_count((int)__builtin_amdgcn_ubfe(b[j], (uint32_t)hist_shift, 31u - (uint32_t)hist_shift) - hist_lo);
                } else {
                    // double: the top 11 bits of the 63-bit magnitude are the exponent field (first level
                    // of wt64_abs_median's select)
                    const unsigned long long b[2] = {(unsigned long long)__double_as_longlong(d0.x),
                                                     (unsigned long long)__double_as_longlong(d0.y)};
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (FAST || x + j < g.W) hist_count((int)((b[j] & 0x7fffffffffffffffull) >> hist_shift) - hist_lo);
                }
            }
        }
        if constexpr (NS == 1) {
            if (esc) wt_bstore4v<WT_FUSED_C_AUX>(rc, at(LAGC), n0);
        }
        V d1 = zero, d2 = zero, d3 = zero, n1 = zero, n2 = zero, n3 = zero;
        if constexpr (NS > 1) {
            if (eh1) {
                if constexpr (WA) n1 = wt_hfilter_dpp<K, ((D << A1) <= 4 ? (D << A1) : 4)>(v1);
                else n1 = wt_hfilter_lds<T, K, (D << A1), NL>(buf[A1], gl, v1);
                d1 = f4_sub(cen1, n1);
            }
            if (es1) wt_bstore4v<WT_FUSED_W_AUX>(rw1, at(LAG1), d1);
            if constexpr (NS == 2) {
                if (esc) wt_bstore4v<WT_FUSED_C_AUX>(rc, at(LAGC), n1);
            }
            if constexpr (NS > 2) {
                if (eh2) {
                    if constexpr (WA) n2 = wt_hfilter_dpp<K, ((D << A2) <= 4 ? (D << A2) : 4)>(v2);
                    else n2 = wt_hfilter_lds<T, K, (D << A2), NL>(buf[A2], gl, v2);
                    d2 = f4_sub(cen2, n2);
                }
                if (es2) wt_bstore4v<WT_FUSED_W_AUX>(rw2, at(LAG2), d2);
                if constexpr (NS == 3) {
                    if (esc) wt_bstore4v<WT_FUSED_C_AUX>(rc, at(LAGC), n2);
                }
                if constexpr (NS > 3) {
                    if (eh3) {
                        n3 = wt_hfilter_lds<T, K, (D << A3), NL>(buf[A3], gl, v3);
                        d3 = f4_sub(cen3, n3);
                    }
                    if (es3) wt_bstore4v<WT_FUSED_W_AUX>(rw3, at(LAG3), d3);
                    if (esc) wt_bstore4v<WT_FUSED_C_AUX>(rc, at(LAGC), n3);
                    c3 = n2;
                }
            }
            c2 = n1;
        }
        if constexpr (SUM) {
            // plane-order sum of image row rho: ((p_in + w_s0) + w_s0+1) + w_s0+2 (+ c): the
            // partial sum of a row is parked in the ring until the next scale's detail row of the
            // same image row appears (G1, then G2 steps later)
            V s = PIN ? f4_add(pin_cur, d0) : d0;                       // row t - LAG0
            if constexpr (NS > 1) {
                V s1 = f4_add(old1, d1);                                  // row t - LAG1
                V s2 = s1;
                if constexpr (NS > 2) s2 = f4_add(old2, d2);              // row t - LAG2
                // (a ring slot written at step k is read when the next scale's detail row of the same
                //  image row comes out: needed from the step at which that row is a stored one)
                if (es0) ring[i1 * (NV + 1) + li] = s;
                if constexpr (NS > 2) {
                    if (es1) ring[(G1 + i2) * (NV + 1) + li] = s1;
                }
                if constexpr (NS > 3) {
                    if (es2) ring[(G1 + G2 + i3) * (NV + 1) + li] = s2;
                    s = f4_add(old3, d3);                                 // row t - LAG3
                } else {
                    s = s2;
                }
                i1 = (i1 + 1 == G1) ? 0 : i1 + 1;
                if constexpr (NS > 2) i2 = (i2 + 1 == G2) ? 0 : i2 + 1;
                if constexpr (NS > 3) i3 = (i3 + 1 == G3) ? 0 : i3 + 1;
            }
            if constexpr (ACC == 2) s = f4_add(s, NS == 1 ? n0 : (NS == 2 ? n1 : (NS == 3 ? n2 : n3)));
            // the finished reconstruction is a write-once stream; an intermediate sum is re-read
            // by the next pass
            if (esc) wt_bstore4v<(ACC == 2 ? WT_FUSED_R_AUX : WT_FUSED_P_AUX)>(rp, at(LAGC), s);
        }
        c1 = n0;
        koff += step_bytes;
#ifndef WT_FUSED_NO_SCHEDBAR
        // Keep the scheduler from interleaving consecutive steps: with the branch-free FAST loads
        // the unrolled body is one basic block, and free motion across steps costs 30 more VGPRs
        // (spills) and turns every wait into vmcnt(0).
        if constexpr (FAST) __builtin_amdgcn_sched_barrier(0);
#endif
    };

    // The register rotation fixes the unroll factor at U steps, not the trip count: leave the
    // body at the last step that stores anything (a loop EXIT, not a skipped step - nothing
    // rejoins inside the loop, so the vmcnt bookkeeping of the steps stays exact).  S = 147 at
    // 8192^2 would otherwise march 192 steps instead of 177.
#ifndef WT_FUSED_NO_EARLY_EXIT
    const int nexact = (r1 - r0) + LAT + LAT_IN;
#else
    const int nexact = nsteps;
#endif
    int kb0 = 0;
    if constexpr (PRO > 0) {
        // the peeled prologue: PRO / U copies of the unrolled body with constant step numbers
        // (compile-time recursion: the step number must be a constant in every copy; as a `#pragma
        // unroll` loop around the early exit the body is NOT unrolled - the register window would
        // land in scratch memory)
        auto prologue = [&](auto self, auto ic) -> bool {
            constexpr int KK = decltype(ic)::value;
            if constexpr (KK < PRO) {
                if (KK >= nexact) return true;               // a chunk shorter than the prologue
                step(KK / U * U, KK % U, std::true_type{});
                return self(self, std::integral_constant<int, KK + 1>{});
            } else {
                return false;
            }
        };
        if (prologue(prologue, std::integral_constant<int, 0>{})) goto done;
        kb0 = PRO;
    } else {
        // (no prologue: NS = 1, or -DWT_FUSED_NO_PROLOGUE)
        // The compiler sizes every `s_waitcnt vmcnt(N)` of the loop from the FEWEST vector-memory
        // operations that can lie between a prefetch and its use on any path into that point - and on
        // the path from here the PD prefetches would be back to back, while in the steady state a
        // step's stores sit between them.  Without the padding below the first PD steps of every trip
        // through the unrolled body wait with vmcnt(2..15), i.e. for the STORES of the previous steps
        // to be acknowledged (once per U steps the wave drains its store queue).  Issue as many parked
        // stores (out-of-range offset: dropped by the range check, no memory traffic) as the steady
        // state has behind the prefetches, so that every wait in the loop becomes vmcnt(~PD*ops/step).
#ifndef WT_FUSED_NO_VMPAD
        constexpr int ST = NS + 1 + (SUM ? 1 : 0);          // stores per step
#pragma unroll
        for (int i = 0; i < PD * ST; ++i) wt_bstore4v<0>(rc, WT_FUSED_PARKED + 16u * i, zero);   // distinct: not merged
#endif
    }
    for (int kb = kb0; kb < nsteps; kb += U) {
#pragma unroll
        for (int kk = 0; kk < U; ++kk) {
            if (kb + kk >= nexact) goto done;
            step(kb, kk, std::false_type{});
        }
    }
done:;
    if constexpr (HIST) {
        if (hist_below) atomicAdd(&lh[0], (uint32_t)hist_below);
        if (hist_out - hist_below) atomicAdd(&lh[WT_HIST_BINS - 1], (uint32_t)(hist_out - hist_below));
        __syncthreads();
        for (int i = threadIdx.x; i < WT_HIST_BINS; i += NL)
            if (lh[i]) atomicAdd(&a.hist[i], lh[i]);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// S = element type (float: wt_plan, double: wt_plan64 - both carry `ctx` and the geometry `g`)
template <typename T, int K, int NS, int D, int NW, int PD, int ACC, typename PLAN>
static int wt_fused_launch_t(PLAN *p, const FusedArgsT<T> &base, const char *name, const FusedRows &rows)
{
    constexpr int PX = WtVec<T>::PX;
    constexpr int ALIGN_PX = 128 / (int)sizeof(T);           // strips start on 128-byte lines

    constexpr int hw = K / 2;
    constexpr int LAT = hw * ((1 << NS) - 1) + (NS - 1);
    constexpr int HX = (hw * ((1 << NS) - 1) * D + ALIGN_PX - 1) / ALIGN_PX * ALIGN_PX;
    constexpr int NL = NW * 64;
    constexpr int VXMAX = (WT_FUSED_WA && D == 1 && PX == 4) ? NW * 56 * 4     // 56 storing lanes per wave
                                                             : NL * PX - 2 * HX; // widest valid strip per WG
    constexpr int UMAX = (K - 1) << (NS - 1);
    static_assert(VXMAX >= 2 * ALIGN_PX, "workgroup too narrow for this halo");
    const Geo &g = p->g;
    FusedArgsT<T> a = base;
    const int W4 = (g.W + PX - 1) / PX * PX;
    const int nx = (W4 + VXMAX / ALIGN_PX * ALIGN_PX - 1) / (VXMAX / ALIGN_PX * ALIGN_PX);
    a.Vx = std::min(VXMAX / ALIGN_PX * ALIGN_PX, ((W4 + nx - 1) / nx + ALIGN_PX - 1) / ALIGN_PX * ALIGN_PX);   // balanced, 128-B aligned
    if ((int64_t)a.Vx * nx < W4) WT_FAIL("fused pass: strip sizing failed");
    const int phases = std::min(D, g.nrows);
    int nranges = rows.n ? rows.n : 1, span_rows = 0;
    for (int i = 0; i < 2; ++i) {
        a.rlo[i] = rows.n ? (i < rows.n ? rows.lo[i] : 0) : 0;
        a.rhi[i] = rows.n ? (i < rows.n ? rows.hi[i] : 0) : (i == 0 ? g.nrows : 0);
        if (a.rlo[i] < 0 || a.rhi[i] > g.nrows || a.rlo[i] > a.rhi[i]) WT_FAIL("fused pass: bad row range [%d,%d)", a.rlo[i], a.rhi[i]);
        span_rows = std::max(span_rows, a.rhi[i] - a.rlo[i]);
    }
    if (span_rows == 0) return 0;
    const int n_max = (span_rows + D - 1) / D;           // longest chain of a range
    // Chunking: the resident capacity is `slots` workgroups (256 CUs x workgroups per CU) and a
    // workgroup's cost is its S stored rows plus the 2*LAT warm-up rows.  Pick the chunk count
    // that minimises (dispatch rounds) x (rows per workgroup): usually ONE round with every
    // slot filled; when the x-strips x phases alone under-fill the chip (tall narrow-ish strips:
    // 160 workgroups for 256 CUs at 32768 columns) a few shorter chunks in two rounds win.
    // (the single-scale passes are light - 92 VGPRs, 16 KB of LDS - two 8-wave workgroups per CU)
    const int wg_per_cu = NW == 4 ? WT_FUSED_WG4_PER_CU(K, NS, ACC) : (NS == 1 ? 2 : std::max(1, 8 / NW));
    static const int rounds_env = getenv("WT_FUSED_ROUNDS") ? std::max(1, atoi(getenv("WT_FUSED_ROUNDS"))) : 0;
    const int64_t nbase = (int64_t)nx * phases * nranges;
    int chunks = 1, S = n_max;
    double best = 1e300;
    auto search = [&](int slots) {
        chunks = 1; S = n_max; best = 1e300;
        for (int c = 1; c <= 4096 && c <= n_max; ++c) {
            const int Sc = (n_max + c - 1) / c;
            const int cc = (n_max + Sc - 1) / Sc;                    // chunks actually needed
            const int64_t rounds = (nbase * cc + slots - 1) / slots;
            // the kernel addresses the rows of a chunk with 31-bit byte offsets
            if ((int64_t)(Sc + 2 * LAT + 2 * UMAX + 1) * D * g.P * (int64_t)sizeof(T) >= ((int64_t)1 << 31)) continue;
            // more than one round: keep the warm-up <= ~50 % of a chunk.  A grid that fits in one
            // round anyway (small images: the chip is not full) is latency-bound by the steps of
            // ONE workgroup, so shorter chunks win even if most of their steps are warm-up.
            if (c > 1 && Sc < std::min(n_max, rounds > 1 ? 2 * LAT : 4)) break;
            if (rounds_env && rounds > rounds_env) break;
            const double cost = (double)rounds * (Sc + 2 * LAT + 8);
            if (cost < best) { best = cost; chunks = cc; S = Sc; }
        }
    };
    const int cus = p->ctx->num_cus;
    search(std::max(64, (cus - rows.reserve) * wg_per_cu));
    // Large images, D = 1: ONE workgroup per CU with chunks twice as long.  The pass is bound by
    // its memory pattern, not by latency (section 3.1 of DESIGN.md), so the second workgroup per CU
    // buys nothing, while half as many chunks halve the warm-up share and the number of isolated
    // write fronts: 8192^2 0.297 -> 0.277 ms, 6144^2 -8 %, 12288^2 -8 %, 16384^2 -5 %
    // (profiles/r02_f_chunks.txt).  At 4096^2 (chunks of 81 rows) it is 3 % slower: the planes sit in
    // the Infinity Cache there and latency matters again - hence the threshold on the chunk length.
    static const int wpc1_env = getenv("WT_FUSED_WPC1") ? atoi(getenv("WT_FUSED_WPC1")) : -1;   // -1 auto, 0 off, 1 force
    if (D == 1 && NW == 4 && wg_per_cu > 1 && wpc1_env != 0) {
        const int c2 = chunks, S2 = S;
        const double b2 = best;
        // (an interior launch beside a halo exchange, rows.reserve > 0: with one 4-wave workgroup
        // per CU half of every CU's register file and LDS stays free for the RCCL kernels, so no CU
        // is set aside - setting 16 aside costs a whole chunk per strip, 7-10 % of the pass at
        // 32768 columns: 36 strips x 7 chunks = 252 workgroups fit 256 CUs, 6 chunks do not fill 240)
        search(std::max(64, cus));
        if (best == 1e300 || (wpc1_env < 0 && S < 128)) { chunks = c2; S = S2; best = b2; }
    }
    if (best == 1e300) WT_FAIL("fused pass: no chunking keeps a chunk's byte span below 2 GiB");
    static const int chunks_env = getenv("WT_FUSED_CHUNKS") ? atoi(getenv("WT_FUSED_CHUNKS")) : 0;   // experiments (D = 1 passes)
    if (D == 1 && chunks_env > 0 && chunks_env <= n_max) {
        S = (n_max + chunks_env - 1) / chunks_env;
        chunks = (n_max + S - 1) / S;
    }
    a.S = S;
    a.chunks = chunks;
    static const int dbg = getenv("WT_FUSED_DEBUG") ? atoi(getenv("WT_FUSED_DEBUG")) : 0;
    a.debug = dbg;
    const int64_t gy = (int64_t)D * chunks;
    if (gy > 65535) WT_FAIL("fused pass: grid too large");
    dim3 grid(nx, (unsigned)gy, nranges), block(NL);
    // (the two parts of a split pass are timed under their own names: bench.py reports per-pass
    //  exchange / interior / edge times of the multi-GPU schedule)
    const std::string pname = std::string(name) + (rows.part == 1 ? "/interior" : rows.part == 2 ? "/edge" : "");
    ProfScope ps(p->ctx, pname.c_str());
    // fast addressing: a group outside the image is a group inside it read backwards (any width since round 6: the one
    // group that straddles the right border is a swizzle of its own four pixels) and no index reflects twice.  (The
    // riding histogram counts whole groups: it keeps the generic addressing for widths the groups do not divide.)
    const bool fast = g_opt_fused_fast && (g.W % PX == 0 || ACC != 3) && g.W >= HX && g.W >= 2 * PX &&
                      g.H >= D * (hw * ((1 << NS) - 1) + 1);
    if (fast) hipLaunchKernelGGL((wt_fused_kernel<T, K, NS, D, NW, PD, ACC, true>), grid, block, 0, p->ctx->stream, a);
    else hipLaunchKernelGGL((wt_fused_kernel<T, K, NS, D, NW, PD, ACC, false>), grid, block, 0, p->ctx->stream, a);
    WT_HIP(hipGetLastError());
    return 0;
}

// Tuned on MI355X (8192^2, B3): the D = 1 pass prefers 4-wave workgroups (2 resident per CU),
// the D = 8 and D = 64 passes 8-wave workgroups (x halo 256 of 2048 px instead of 256 of 1024);
// 4 rows of prefetch.  acc: 0 = plain pass, 1 = also carry the plane sum (p_in -> p_out),
// 2 = last pass of a sum (adds the smooth plane, streaming store).
template <int K, int ACC>
static int wt_fused_dispatch_acc(wt_plan *p, const FusedArgs &a, int s0, int ns, const FusedRows &rows)
{
    typedef float T;
    static const char *names[4][7] = {
        {"wt_fused<d1x3>", "wt_fused<d1x2>", "wt_fused<d8x3>", "wt_fused<d8x2>", "wt_fused<d64x2>", "wt_fused<d1x4>", "wt_fused<d16x4>"},
        {"wt_fused_acc<d1x3>", "wt_fused_acc<d1x2>", "wt_fused_acc<d8x3>", "wt_fused_acc<d8x2>", "wt_fused_acc<d64x2>", "wt_fused_acc<d1x4>",
         "wt_fused_acc<d16x4>"},
        {"wt_fused_sum<d1x3>", "wt_fused_sum<d1x2>", "wt_fused_sum<d8x3>", "wt_fused_sum<d8x2>", "wt_fused_sum<d64x2>", "wt_fused_sum<d1x4>",
         "wt_fused_sum<d16x4>"},
        {"wt_fused_hist<d1x3>", "wt_fused_hist<d1x2>", "", "", "", "wt_fused_hist<d1x4>", ""}};
    // 3-tap family: FOUR scales per pass (register window 2 * 15 float4): L = 8 is two passes,
    // (0,4) at D = 1 and (4,4) at D = 16 (x halo 240 px: 8-wave workgroups; the accumulate variants
    // take 7 waves, their delay rings of 3 + 5 + 9 rows would not fit the LDS with 8)
    if constexpr (K == 3) {
        if (s0 == 0 && ns == 4) return wt_fused_launch_t<T, K, 4, 1, 4, 4, ACC>(p, a, names[ACC][5], rows);
        if constexpr (ACC != 3) {
            if (s0 == 4 && ns == 4)
                return wt_fused_launch_t<T, K, 4, 16, (ACC == 0 ? 8 : 7), (ACC == 0 ? 4 : 2), ACC>(p, a, names[ACC][6], rows);   // (accumulate: 2 rows ahead, 256 VGPRs)
        }
    }
    if (s0 == 0 && ns == 3) return wt_fused_launch_t<T, K, 3, 1, 4, 4, ACC>(p, a, names[ACC][0], rows);
    if (s0 == 0 && ns == 2) return wt_fused_launch_t<T, K, 2, 1, 4, 4, ACC>(p, a, names[ACC][1], rows);
    if constexpr (ACC != 3) {
        if (s0 == 3 && ns == 3) return wt_fused_launch_t<T, K, 3, 8, 8, 4, ACC>(p, a, names[ACC][2], rows);
        if (s0 == 3 && ns == 2) return wt_fused_launch_t<T, K, 2, 8, 8, 4, ACC>(p, a, names[ACC][3], rows);
        // D = 64 (scales 6-7): taps are 16 / 32 lanes apart
        if (s0 == 6 && ns == 2) return wt_fused_launch_t<T, K, 2, 64, 8, 4, ACC>(p, a, names[ACC][4], rows);
    }
    // Single-scale passes that END a schedule ((3,1) of 4 scales, (6,1) of 7): with them every pass
    // of such a schedule can carry the plane sum (wt_decompose_sum, the interleaved denoise).  The
    // plain variant exists so that wt_decompose writes the same bits as wt_decompose_sum (the fused
    // passes filter rows first, the per-scale kernels columns first: last-bit differences).
    if constexpr (ACC != 3) {
        static const char *names1[3][2] = {{"wt_fused<d8x1>", "wt_fused<d64x1>"}, {"wt_fused_acc<d8x1>", "wt_fused_acc<d64x1>"},
                                           {"wt_fused_sum<d8x1>", "wt_fused_sum<d64x1>"}};
        if (s0 == 3 && ns == 1) return wt_fused_launch_t<T, K, 1, 8, 8, (K == 5 ? 4 : 2), ACC>(p, a, names1[ACC][0], rows);
        if (s0 == 6 && ns == 1) return wt_fused_launch_t<T, K, 1, 64, 8, (K == 5 ? 4 : 2), ACC>(p, a, names1[ACC][1], rows);
    }
    WT_FAIL("fused pass (first scale %d, %d scales) is not built", s0, ns);
}

// the float64 passes (wt_plan64; called from wt_f64.h through wt_fused_tu.hip)
template <int K, int ACC>
static int wt_fused64_dispatch_acc(wt_plan64 *p, const FusedArgsT<double> &a, int s0, int ns, const FusedRows &rows)
{
    typedef double T;
    static const char *pre[4] = {"wt64_fused", "wt64_fused_acc", "wt64_fused_sum", "wt64_fused_hist"};
    static char names[4][16][32];
    auto nm = [&](int slot, const char *tag) -> const char * {
        if (!names[ACC][slot][0]) snprintf(names[ACC][slot], sizeof names[ACC][slot], "%s<%s>", pre[ACC], tag);
        return names[ACC][slot];
    };
    // Workgroup shapes as the float passes: 4 waves at D = 1 (512 pixels per row step), 8 waves for
    // the dilated passes (1024 pixels, x halo 112 / 48 / 192 px per side) - the four-scale accumulate
    // variants included: a double2 lane owns 2 pixels, so the 240-pixel halo leaves 272 storing lanes
    // and the delay rings (17 rows x 273 x 16 B) fit beside the row buffers (140 KB of LDS), where the
    // float variant (392 storing lanes) has to drop to 7 waves.  8 instead of 7 waves: 0.86 -> 0.75 ms
    // for the (4,4) pass that carries the sum at 8192^2.
    if constexpr (K == 3) {
        if (s0 == 0 && ns == 4) return wt_fused_launch_t<T, K, 4, 1, 4, 4, ACC>(p, a, nm(7, "d1x4"), rows);
        if constexpr (ACC != 3) {
            if (s0 == 4 && ns == 4) return wt_fused_launch_t<T, K, 4, 16, 8, (ACC == 0 ? 4 : 2), ACC>(p, a, nm(8, "d16x4"), rows);
        }
    }
    if (s0 == 0 && ns == 3) return wt_fused_launch_t<T, K, 3, 1, 4, 4, ACC>(p, a, nm(0, "d1x3"), rows);
    if (s0 == 0 && ns == 2) return wt_fused_launch_t<T, K, 2, 1, 4, 4, ACC>(p, a, nm(1, "d1x2"), rows);
    if constexpr (ACC != 3) {      // (the histogram variant exists for the first pass only)
        if (s0 == 3 && ns == 3) return wt_fused_launch_t<T, K, 3, 8, 8, 4, ACC>(p, a, nm(2, "d8x3"), rows);
        if (s0 == 3 && ns == 2) return wt_fused_launch_t<T, K, 2, 8, 8, 4, ACC>(p, a, nm(3, "d8x2"), rows);
        if (s0 == 6 && ns == 2) return wt_fused_launch_t<T, K, 2, 64, 8, 4, ACC>(p, a, nm(4, "d64x2"), rows);
        if (s0 == 3 && ns == 1) return wt_fused_launch_t<T, K, 1, 8, 8, (K == 5 ? 4 : 2), ACC>(p, a, nm(5, "d8x1"), rows);
        if (s0 == 6 && ns == 1) return wt_fused_launch_t<T, K, 1, 64, 8, (K == 5 ? 4 : 2), ACC>(p, a, nm(6, "d64x1"), rows);
    }
    WT_FAIL("float64 fused pass (first scale %d, %d scales) is not built", s0, ns);
}

