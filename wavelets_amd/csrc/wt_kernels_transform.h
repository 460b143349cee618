// Kernels of the transform unit (wt_transform.hip): the float32 bilateral march in its two forms and the column
// pass of the run-time-tap filter.  (The chain / lattice / row kernels are wt_stencil.h, the fused passes
// wt_fused.h.)  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "wt_internal.h"
#include "wt_device.h"
#include "wt_stencil.h"
#include "wt_kernels_common.h"

// ---------------------------------------------------------------------------------------------
// K10  bilateral (range-weighted) dilated convolution - watroo/wavelets.py:74-105
//   out = (k_c I + sum_t k_t e_t I_t) / (k_c + sum_t k_t e_t),
//   e_t = exp(-((I - I_t)^2) / var / 2)                        (numexpr expression, :97)
// Full K x K tap set (not separable): K*K-1 exponentials per pixel.  Per tap the weight is ONE v_exp_f32:
//   k_t * exp(-d^2/(2 var)) = 2^( d^2 * (-log2(e)/(2 var)) + log2(k_t) )
// with the per-pixel factor -log2(e)/(2 var) formed once (one division per pixel instead of
// one per tap).  fp32 rounding differs from the reference's exp()/divide sequence by a few
// ulp of the weight - inside the stated bilateral tolerance.
// Work decomposition is the chain march of K1: a thread owns a group of columns and one polyphase row
// chain, and keeps the K x K (dilated) neighbourhood rows in a register window that advances one
// chain step per iteration - every input row is fetched once per chain (K coalesced
// loads at x + j*d, L2-served) instead of once per output row, which is what makes the large
// dilations of wow() (d up to 1024, where a tile has no spatial reuse) HBM-neutral.
// (Until round 5 a four-pixel-per-thread form of this kernel was kept beside the two-pixel one as its bitwise
//  cross-check; it lost every timing since round 3 and went in round 6 - the cross-check is now the generic
//  load path of the kernel below, option "bilateral_paired" = 0.)
// ---------------------------------------------------------------------------------------------
// K10b  TWO pixels per thread (any dilation): a K x K float2 window, 92 VGPRs, 5 waves per SIMD.
//
// Round 6: the row that enters the window is fetched with raw BUFFER loads (SGPR descriptor of the row +
// one 32-bit lane offset per operand, 2 K four-byte loads per row) instead of per-lane branches between one
// 8-byte and two 4-byte flat loads.  The branches cost nothing by themselves, but loads inside divergent
// regions make the number of loads in flight path-dependent, so the compiler's wait-count pass fell back to
// `s_waitcnt vmcnt(0)` at the join - in front of the tap loop: the "prefetched" row was waited for BEFORE the
// 24 taps it was meant to hide behind, and the VALU idled whenever the four waves of a SIMD sat in that wait
// together (0.78 issue-busy).  With straight-line loads the wait is counted exactly and lands where the row is
// first used, one whole step later.  For the same reason the variance source is a template parameter (the
// plane form loads through the same descriptors), the LDS ring slots follow the unroll phase (immediate
// offsets), and only the border rules the launch code admits (0, 1) are compiled in.
// ---------------------------------------------------------------------------------------------
typedef unsigned int wt_su2 __attribute__((ext_vector_type(2)));
typedef float wt_sf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wt_store2(float *row, int x, int P, float2 v)
{
    const wt_sf2 t = {v.x, v.y};
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(wt_su2, t), wt_row_rsrc(row, P), (unsigned)x * 4u, 0, 0);
}
template <int K, bool INLINE_VAR, bool PAIRED>
__global__ __launch_bounds__(256) void wt_bilateral2_kernel(ChainArgs a)
{
    constexpr int hw = K / 2;
    const Geo g = a.g;
    int bx, by;
    wt_xcd_remap(bx, by);
    // the waves of a workgroup sit SIDE BY SIDE on the same chain (round 6): a workgroup reads and writes
    // blockDim.y * 512 contiguous bytes of one row per step.  (Until round 5 each wave had a chain of its own and the
    // chip kept ~5 000 rows open with 512-byte accesses: HBM pages, not the VALU, set the kernel's pace.)
    const int x = ((bx * (int)blockDim.y + (int)threadIdx.y) * 64 + threadIdx.x) * 2;
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

    // Operand columns do not depend on the row: pixel pair x + (j - hw) d, reflected per pixel at the image border.
    // PAIRED (the symmetric border of the whole image, Geo::border 0): two neighbouring indices reflect to the same
    // or to neighbouring pixels whatever the number of bounces, so the pair is ONE 8-byte load at the lower of the
    // two (unaligned for odd operands of d = 1: buffer loads need 4-byte alignment only) and at most a swap -
    // (v0, v1), (v1, v0), (v0, v0) or (v1, v1) by two selects, which only waves that touch the left or right image
    // border execute.  Otherwise (reflection inside polyphase components) 2 K four-byte loads.
    unsigned xa[K], xb[K];
    bool sa[K], sb[K];
    bool odd = false;
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int xo = x + (j - hw) * d;
        const int ia = wt_refl_01(xo, g.W, d, g.border), ib = wt_refl_01(xo + 1, g.W, d, g.border);
        if constexpr (PAIRED) {
            const int lo = max(min(min(ia, ib), g.P - 2), 0);
            xa[j] = 4u * (unsigned)lo;
            sa[j] = ia != lo;
            sb[j] = ib != lo;
            odd = odd || sa[j] || !sb[j];
        } else {
            xa[j] = 4u * (unsigned)ia;
            xb[j] = 4u * (unsigned)ib;
        }
    }
    const bool plain = !PAIRED || __builtin_amdgcn_ballot_w64(odd) == 0;     // wave-uniform: no operand of this wave is reflected
    float2 win[K][K];
    auto load_win_row = [&](int r, float2 (&dst)[K]) {
        const int ry = wt_refl_01(gy0 + d * r, g.H, d, g.border);
        const __amdgpu_buffer_rsrc_t rs = wt_row_rsrc(a.in + (int64_t)(ry - g.row0) * g.P, g.P);
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if constexpr (PAIRED) {
                const wt_sf2 v = __builtin_bit_cast(wt_sf2, __builtin_amdgcn_raw_buffer_load_b64(rs, xa[j], 0, 0));
                dst[j] = make_float2(v.x, v.y);
            } else {
                dst[j] = make_float2(__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, xa[j], 0, 0)),
                                     __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, xb[j], 0, 0)));
            }
        }
    };
    // the swap of a loaded row's reflected operands (behind a wave-uniform branch that holds no memory operation:
    // the compiler's wait counts stay exact)
    auto fix_row = [&](float2 (&row)[K]) {
        if constexpr (PAIRED) {
            if (!plain) {
#pragma unroll
                for (int j = 0; j < K; ++j) row[j] = make_float2(sa[j] ? row[j].y : row[j].x, sb[j] ? row[j].y : row[j].x);
            }
        }
    };
#pragma unroll
    for (int i = 0; i < K; ++i) load_win_row(r0 - hw + i, win[i]);
#pragma unroll
    for (int i = 0; i < K; ++i) fix_row(win[i]);
    float2 nxt[K];

    // In-kernel variance: the row filters (h = row-filtered I, h2 = row-filtered I^2) of a window
    // row are computed ONCE, when the row enters, and parked in a per-thread LDS ring of K slots
    // (no other thread touches them: no barrier); every step reads the K pairs for the column
    // filter instead of filtering all K rows again (4/5 of that arithmetic, ~20 % of the kernel's
    // VALU work; at 4 waves per SIMD the kernel is VALU-bound).  Same operations in the same
    // order as wt_hrow_filter<MODE_VAR> + WtVert: bit-identical to the separate variance pass.
    __shared__ float2 hring[INLINE_VAR ? K : 1][2][256];
    const int tid = threadIdx.y * 64 + threadIdx.x;
    auto row_filters = [&](const float2 (&wr)[K], float2 &h, float2 &h2) {
        float hh[2], hh2[2];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float v[2] = {wr[j].x, wr[j].y};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float sq = v[k] * v[k];
                hh[k] = (j == 0) ? wt_tap<K>(0) * v[k] : fmaf(wt_tap<K>(j), v[k], hh[k]);
                hh2[k] = (j == 0) ? wt_tap<K>(0) * sq : fmaf(wt_tap<K>(j), sq, hh2[k]);
            }
        }
        h = make_float2(hh[0], hh[1]);
        h2 = make_float2(hh2[0], hh2[1]);
    };
    if constexpr (INLINE_VAR) {
#pragma unroll
        for (int i = 0; i < K - 1; ++i) {
            float2 h, h2;
            row_filters(win[i], h, h2);
            hring[i][0][tid] = h;
            hring[i][1][tid] = h2;
        }
    }

    const float kc = wt_tap<K>(hw) * wt_tap<K>(hw);
    // One step of the march.  The window does NOT slide through the registers (K * K 8-byte moves per
    // row, ~10 % of the kernel's vector instructions): the row loop is unrolled K times and in phase U
    // window row i lives in slot (i + U) % K - the entering row replaces the row that left (K moves) - and
    // its row filters in ring slot (i + U) % K likewise.
    // Same operations in the same order in every phase: identical bits.
    auto step = [&](const int r, auto utag) {
        constexpr int U = decltype(utag)::value;
        load_win_row(min(r + 1, r1 - 1) + hw, nxt);      // software prefetch of the entering row
        const int64_t roff = (int64_t)(q + d * r) * g.P;
        const float I[2] = {win[(hw + U) % K][hw].x, win[(hw + U) % K][hw].y};
        float vv[2];
        if constexpr (INLINE_VAR) {
            float2 hn, h2n;
            row_filters(win[(K - 1 + U) % K], hn, h2n);  // the row that entered the window
            hring[(K - 1 + U) % K][0][tid] = hn;
            hring[(K - 1 + U) % K][1][tid] = h2n;
            float m[2], p[2];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                float2 h, h2;
                if (i < K - 1) {
                    h = hring[(i + U) % K][0][tid];
                    h2 = hring[(i + U) % K][1][tid];
                } else {
                    h = hn;
                    h2 = h2n;
                }
                const float hk[2] = {h.x, h.y}, h2k[2] = {h2.x, h2.y};
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    m[k] = (i == 0) ? wt_tap<K>(0) * hk[k] : fmaf(wt_tap<K>(i), hk[k], m[k]);
                    p[k] = (i == 0) ? wt_tap<K>(0) * h2k[k] : fmaf(wt_tap<K>(i), h2k[k], p[k]);
                }
            }
            vv[0] = wt_var_point(p[0], m[0], a.f1, a.f2, 0);
            vv[1] = wt_var_point(p[1], m[1], a.f1, a.f2, 0);
        } else {
            // variance plane: both pixels through the row's descriptor (the second column clamped into the
            // row: a lane whose second pixel is past the image stores nothing for it)
            const __amdgpu_buffer_rsrc_t rv = wt_row_rsrc(a.aux + roff, g.P);
            vv[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, 4u * (unsigned)x, 0, 0));
            vv[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rv, 4u * (unsigned)min(x + 1, g.W - 1), 0, 0));
        }
        // The two pixels of a thread are a register PAIR throughout the tap loop: difference,
        // square, exponent (one v_pk_fma with the tap's log2 weight as the addend), and the two
        // accumulations are packed-FP32 instructions; only the exponentials are per pixel.
        typedef float wt_p2 __attribute__((ext_vector_type(2)));
        const wt_p2 Iv = {I[0], I[1]};
        wt_p2 norm = {kc, kc};
        wt_p2 acc = kc * Iv;
        const wt_p2 s2 = {wt_div_nr(-0.72134752044448170368f, vv[0]), wt_div_nr(-0.72134752044448170368f, vv[1])};   // -log2(e) / (2 var)
        // taps in the reference order (watroo/wavelets.py:89-91): kernel index (i, j) pairs with the shift
        // (K-1-i-hw, K-1-j-hw) * d
#pragma unroll
        for (int i = 0; i < K; ++i) {
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (i == hw && j == hw) continue;
                const float lk = wt_tap_log2<K>(i) + wt_tap_log2<K>(j);
                const float2 t2 = win[(K - 1 - i + U) % K][K - 1 - j];
                const wt_p2 t = {t2.x, t2.y};
                const wt_p2 diff = Iv - t;
                const wt_p2 ex = __builtin_elementwise_fma(diff * diff, s2, (wt_p2){lk, lk});
                const wt_p2 w = {__builtin_amdgcn_exp2f(ex.x), __builtin_amdgcn_exp2f(ex.y)};
                norm += w;
                acc = __builtin_elementwise_fma(t, w, acc);
            }
        }
        float o[2], ow[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            o[k] = wt_div_nr(acc[k], norm[k]);
            ow[k] = I[k] - o[k];                           // detail plane, wavelets.py:442
        }
        wt_store2(a.out_c + roff, x, g.P, make_float2(o[0], o[1]));
        if (a.out_w) wt_store2(a.out_w + roff, x, g.P, make_float2(ow[0], ow[1]));
        fix_row(nxt);
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

__global__ __launch_bounds__(256) void wt_custom_cols_kernel(const float *tmp, const float *in, float *out_c,
                                                             float *out_w, Geo g, int d, CustomTaps t)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= g.W) return;
    const int hw = t.n / 2;
    for (int y = blockIdx.y; y < g.nrows; y += gridDim.y) {
        float acc = 0.f;
        for (int i = 0; i < t.n; ++i) {
            const int yy = wt_refl_b(g.row0 + y + (i - hw) * d, g.H, d, g.border) - g.row0;
            const float v = tmp[(int64_t)yy * g.P + x];
            acc = i == 0 ? t.k[0] * v : fmaf(t.k[i], v, acc);
        }
        const int64_t o = (int64_t)y * g.P + x;
        if (out_w) out_w[o] = in[o] - acc;
        out_c[o] = acc;
    }
}

