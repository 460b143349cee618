// K10 in float64: the bilateral (range-weighted) dilated convolution of watroo/wavelets.py:74-105 on
// double planes - what AtrousTransform(bilateral=...) / wow(bilateral=...) run for float64 and integer
// (FITS) images, which the reference computes in float64 (wavelets.py:297,319-320).
//
//   out = (k_c I + sum_t k_t e_t I_t) / (k_c + sum_t k_t e_t),   e_t = exp(-((I - I_t)^2) / var / 2)   (:97)
//
// The march of wt_bilateral2_kernel (wt_kernels_transform.h) with ONE pixel per thread: a thread owns a column and
// one chunk of one polyphase row chain and keeps the K x K dilated neighbourhood in a register window of
// K x K doubles (the float kernel's K x K float2), every input row is fetched once per chain through K
// coalesced 8-byte loads at x + j d.  Full K x K tap set (not separable): VALU-bound, K*K - 1 exponentials
// per pixel, and there is no double-precision exponential instruction - per tap
//   k_t exp(-delta^2 / (2 var)) = 2^(delta^2 * (-log2(e) / (2 var)) + log2(k_t))
// with the per-pixel factor formed once: u = fma(delta^2, s / 64, 1 + log2 k_t / 64) clamped to [0, 1] by the FMA's
// own output modifier, the range reduction by the 1.5 * 2^37 trick, 2^(j / 512) from a 512-entry table in LDS, a degree-3
// polynomial, the exponent added as an integer (wt_math64.h, table form).  12 double-precision operations (5 of them
// FMAs) + 4 integer operations + 1 LDS read per tap, ~305 double-precision operations per pixel with the variance
// and the two divisions.  The kernel is bound by its double-precision issue at the clock the chip holds under it
// (DESIGN.md section 3.6): the first version (a degree-10 polynomial, 12 FMAs per tap at the SAME instruction count)
// took 1.20 ms per scale of 8192^2, the 64-entry table 1.08, this one 1.05 (round 6: buffer loads, see below) -
// against 0.34 ms for the float kernel whose taps are packed-FP32 pairs around a hardware v_exp_f32.
// Differences from the reference's operation order (exp of a quotient; IEEE divisions) are a few ulp of
// the weight: the float64 parity bound of the tests is 1e-12 * max|input|.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "wt_stencil.h"

template <int K>
__device__ __forceinline__ constexpr double wt_tap_log2_d(int i)
{
    // log2 of the taps: 1/4, 1/2 (Triangle); 1/16, 1/4, 3/8 (B3spline)
    if (K == 3) return i == 1 ? -1.0 : -2.0;
    return (i == 2) ? -1.4150374992788438 : ((i == 1 || i == 3) ? -2.0 : -4.0);
}

typedef unsigned int wt_du2 __attribute__((ext_vector_type(2)));
// 8-byte store of one pixel through a raw buffer descriptor of the row (see wt_storev)
__device__ __forceinline__ void wt_store1d(double *row, int x, int P, double v)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(wt_du2, v), wt_row_rsrc(row, P), (unsigned)x * 8u, 0, 0);
}

// Round 6 (as in wt_bilateral2_kernel): the entering row comes through raw buffer loads of the row's descriptor and the
// variance source is a template parameter, so no memory operation sits inside a branch and the compiler's wait counts
// are exact (the run-time `inline_var` test put an `s_waitcnt vmcnt(0)` in front of every step's taps); the ring slots
// follow the unroll phase; the waves of a workgroup sit side by side on one chain.
template <int K, bool INLINE_VAR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void wt64_bilateral_march_kernel(ChainArgsT<double> a)
{
    constexpr int hw = K / 2;
#if WT_BIL64_TABLE
    __shared__ double exp2tab[WT_BIL64_TABLE];            // (before any wave leaves: every wave of the workgroup reads it)
#ifdef WT_EXP2T_COMPUTED
    for (int j = threadIdx.y * 64 + threadIdx.x; j < WT_BIL64_TABLE; j += 256) exp2tab[j] = wt_exp2_64((double)j * (1.0 / WT_BIL64_TABLE) - 64.0);
#else
    if (threadIdx.y == 0 && threadIdx.x < WT_BIL64_TABLE) exp2tab[threadIdx.x] = WT_EXP2T_T[threadIdx.x];
#endif
    __syncthreads();
#endif
    const Geo g = a.g;
    int bx, by;
    wt_xcd_remap(bx, by);
    const int x = (bx * (int)blockDim.y + (int)threadIdx.y) * 64 + threadIdx.x;
    if (x >= g.W) return;
    const int item = __builtin_amdgcn_readfirstlane(by);   // the chain item (phase, chunk) of this workgroup: scalar
    const int d = a.d;
    const int q = item % d;
    const int c = item / d;
    if (c >= a.chunks || q >= g.nrows) return;
    const int n_q = (g.nrows - q + d - 1) / d;
    const int r0 = c * a.S;
    const int r1 = min(r0 + a.S, n_q);
    if (r0 >= r1) return;
    const int gy0 = g.row0 + q;

    // operand columns do not depend on the row: pixel x + (j - hw) d, reflected at the image border
    unsigned xo[K];                                       // byte offsets into a row
#pragma unroll
    for (int j = 0; j < K; ++j) xo[j] = (unsigned)wt_refl_01(x + (j - hw) * d, g.W, d, g.border) * 8u;
    double win[K][K];
    auto load_win_row = [&](int r, double (&dst)[K]) {
        const int ry = wt_refl_01(gy0 + d * r, g.H, d, g.border);
        const __amdgpu_buffer_rsrc_t rs = wt_row_rsrc(a.in + (int64_t)(ry - g.row0) * g.P, g.P);
#pragma unroll
        for (int j = 0; j < K; ++j) dst[j] = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, xo[j], 0, 0));
    };
#pragma unroll
    for (int i = 0; i < K; ++i) load_win_row(r0 - hw + i, win[i]);
    double nxt[K];

    // In-kernel variance: the row filters (h = row-filtered I, h2 = row-filtered I^2) of a window row are
    // computed ONCE, when the row enters, and parked in a per-thread LDS ring of K slots (no other thread
    // touches them: no barrier); every step reads the K pairs for the column filter.  Same operations in
    // the same order as wt_hrow_filter<MODE_VAR> + WtVert (and as wt64_rows_kernel + the column pass of
    // the float64 engine: FMA chains in tap order): bit-identical to the separate variance pass.
    __shared__ double hring[INLINE_VAR ? K : 1][2][256];
    const int tid = threadIdx.y * 64 + threadIdx.x;

    auto row_filters = [&](const double (&wr)[K], double &h, double &h2) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const double v = wr[j];
            const double sq = v * v;
            h = (j == 0) ? wt_tap_s<K, double>(0) * v : fma(wt_tap_s<K, double>(j), v, h);
            h2 = (j == 0) ? wt_tap_s<K, double>(0) * sq : fma(wt_tap_s<K, double>(j), sq, h2);
        }
    };
    if constexpr (INLINE_VAR) {
#pragma unroll
        for (int i = 0; i < K - 1; ++i) {
            double h, h2;
            row_filters(win[i], h, h2);
            hring[i][0][tid] = h;
            hring[i][1][tid] = h2;
        }
    }

    const double kc = wt_tap_s<K, double>(hw) * wt_tap_s<K, double>(hw);
    // One step of the march.  The window does NOT slide through the registers: the row loop is unrolled K
    // times and in phase U window row i lives in slot (i + U) % K - the entering row replaces the row that
    // left (K moves).  Same operations in the same order in every phase: identical bits.
    auto step = [&](const int r, auto utag) {
        constexpr int U = decltype(utag)::value;
        load_win_row(min(r + 1, r1 - 1) + hw, nxt);      // software prefetch of the entering row
        const int64_t roff = (int64_t)(q + d * r) * g.P;
        const double I = win[(hw + U) % K][hw];
        double vv;
        if constexpr (INLINE_VAR) {
            double hn, h2n;
            row_filters(win[(K - 1 + U) % K], hn, h2n);  // the row that entered the window
            hring[(K - 1 + U) % K][0][tid] = hn;
            hring[(K - 1 + U) % K][1][tid] = h2n;
            double m, p;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                double h, h2;
                if (i < K - 1) {
                    h = hring[(i + U) % K][0][tid];
                    h2 = hring[(i + U) % K][1][tid];
                } else {
                    h = hn;
                    h2 = h2n;
                }
                m = (i == 0) ? wt_tap_s<K, double>(0) * h : fma(wt_tap_s<K, double>(i), h, m);
                p = (i == 0) ? wt_tap_s<K, double>(0) * h2 : fma(wt_tap_s<K, double>(i), h2, p);
            }
            vv = wt_var_point(p, m, a.f1, a.f2, 0);
        } else {
            vv = __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(wt_row_rsrc(a.aux + roff, g.P), 8u * (unsigned)x, 0, 0));
        }
        double norm = kc;
        double acc = kc * I;
        const double s2 = wt_div64(-0.72134752044448170368 / 64.0, vv);   // -log2(e) / (2 var), over 64 (wt_exp2_64_from_u)
        // taps in the reference order (watroo/wavelets.py:89-91): kernel index (i, j) pairs with the shift
        // (K-1-i-hw, K-1-j-hw) * d.  FOUR weights are evaluated in lockstep: one weight is a chain of 18
        // dependent double-precision operations, and compiled tap by tap the kernel ran them one after the
        // other through the same registers (77 % of its issue rate at four waves per SIMD); the accumulation
        // stays in tap order.
        constexpr int NT = K * K - 1, B = 4;
        static_assert(NT % B == 0, "taps come in batches of four");
#if WT_BIL64_TABLE
        const double *C = WT_EXP2T_C;
#pragma unroll
        for (int b0 = 0; b0 < NT; b0 += B) {
            double tv[B], g[B], pw[B], tj[B];
            int e[B];
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const int idx = b0 + k < hw * K + hw ? b0 + k : b0 + k + 1;
                const int i = idx / K, j = idx % K;
                const double lk = 1.0 + (wt_tap_log2_d<K>(i) + wt_tap_log2_d<K>(j)) / 64.0;
                tv[k] = win[(K - 1 - i + U) % K][K - 1 - j];
                const double diff = I - tv[k];
                wt_exp2t_split(fmin(fmax(fma(diff * diff, s2, lk), 0.0), 1.0), g[k], e[k]);
                tj[k] = exp2tab[e[k] & (WT_BIL64_TABLE - 1)];
                pw[k] = C[WT_EXP2T_DEG];
            }
#pragma unroll
            for (int c = WT_EXP2T_DEG - 1; c >= 0; --c) {
#pragma unroll
                for (int k = 0; k < B; ++k) pw[k] = fma(pw[k], g[k], C[c]);
            }
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const double w = wt_exp2t_join(pw[k], tj[k], e[k]);
                norm += w;
                acc = fma(tv[k], w, acc);
            }
        }
#else
        const double *C = WT_EXP2U_C;
#pragma unroll
        for (int b0 = 0; b0 < NT; b0 += B) {
            double tv[B], g[B], pw[B];
            int e[B];
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const int idx = b0 + k < hw * K + hw ? b0 + k : b0 + k + 1;      // (the centre tap is skipped)
                const int i = idx / K, j = idx % K;
                const double lk = 1.0 + (wt_tap_log2_d<K>(i) + wt_tap_log2_d<K>(j)) / 64.0;
                tv[k] = win[(K - 1 - i + U) % K][K - 1 - j];
                const double diff = I - tv[k];
                // u = 1 + (delta^2 s + log2 k_t) / 64 clamped to [0, 1]: the clamp folds into the FMA
                wt_exp2u_split(fmin(fmax(fma(diff * diff, s2, lk), 0.0), 1.0), g[k], e[k]);
                pw[k] = C[10];
            }
#pragma unroll
            for (int c = 9; c >= 0; --c) {
#pragma unroll
                for (int k = 0; k < B; ++k) pw[k] = fma(pw[k], g[k], C[c]);
            }
#pragma unroll
            for (int k = 0; k < B; ++k) {
                const double w = wt_exp2u_join(pw[k], e[k]);
                norm += w;
                acc = fma(tv[k], w, acc);
            }
        }
#endif
        const double o = wt_div64(acc, norm);
        wt_store1d(a.out_c + roff, x, g.P, o);
        if (a.out_w) wt_store1d(a.out_w + roff, x, g.P, I - o);    // detail plane, wavelets.py:442
#pragma unroll
        for (int j = 0; j < K; ++j) win[U][j] = nxt[j];    // slot of the row that left <- the row that entered
    };
    int r = r0;
    while (true) {
        step(r, std::integral_constant<int, 0>{});
        if (++r >= r1) break;
        step(r, std::integral_constant<int, 1>{});
        if (++r >= r1) break;
        step(r, std::integral_constant<int, 2>{});
        if (++r >= r1) break;
        if constexpr (K > 3) {
            step(r, std::integral_constant<int, 3>{});
            if (++r >= r1) break;
            step(r, std::integral_constant<int, 4>{});
            if (++r >= r1) break;
        }
    }
}
