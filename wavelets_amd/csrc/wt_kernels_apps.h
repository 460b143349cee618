// Kernels of the application unit (wt_apps.hip): the pointwise operators (plane sum, thresholds, wow update, gamma
// blend, Anscombe), cubes, the small-PSF correlation of richardson_lucy, the binary / support updates, the exact
// median select and the reductions.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "wt_internal.h"
#include "wt_device.h"
#include "wt_stencil.h"
#include "wt_kernels_common.h"

// ---------------------------------------------------------------------------------------------
// pointwise kernels over the strip's owned rows: rows are contiguous (pitch P), so they are a
// flat float4 range of nrows*P/4 elements.  Grid-stride, 16 B per lane.
// ---------------------------------------------------------------------------------------------
// (WT_MAX_SUM_PLANES: wt_internal.h)
struct SumArgs {
    const float *p[WT_MAX_SUM_PLANES];
    int n;
};

typedef float wt_nt4 __attribute__((ext_vector_type(4)));
// streaming (non-temporal) 16-byte load: planes that are read exactly once should not displace
// L2 / Infinity-Cache lines (measured on MI355X, 7 reads + 1 write: 4.7 -> 6.1 TB/s)
__device__ __forceinline__ float4 wt_ldnt4(const float *p)
{
    const wt_nt4 v = __builtin_nontemporal_load(reinterpret_cast<const wt_nt4 *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ void wt_stnt4(float *p, float4 v)
{
    wt_nt4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<wt_nt4 *>(p));
}
// The reconstruction is a write-once stream too; with default stores its dirty lines are still
// being written back when the next transform's first pass starts (8192^2: that pass 0.36 ->
// 0.32 ms in a frame loop).
#ifndef WT_SUM_NT_STORE
#define WT_SUM_NT_STORE 1
#endif

// K5  np.sum(planes, axis=0): sequential fp32 accumulation in plane order (bit-exact vs numpy).
// One float4 per thread (no grid-stride loop): a large grid of short-lived waves keeps the most
// loads in flight for this 7-reads-1-write stream.
__global__ __launch_bounds__(256) void wt_plane_sum_kernel(SumArgs a, float *out, int64_t n4)
{
#pragma clang fp contract(off)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        float4 acc = wt_ldnt4(a.p[0] + 4 * i);
        for (int k = 1; k < a.n; ++k) {
            const float4 v = wt_ldnt4(a.p[k] + 4 * i);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        if (WT_SUM_NT_STORE) wt_stnt4(out + 4 * i, acc);
        else reinterpret_cast<float4 *>(out)[i] = acc;
    }
}


// K4+K5 fused: dst = sum_k plane_k with the first n_den planes thresholded on the fly
// (plane_k * (wgt_k * significance_k)); optionally writes the thresholded planes back so the
// result is exactly Coefficients.denoise (wavelets.py:145-149) followed by np.sum (utils.py:98)
// in one pass over the planes: saves the read-modify-write of the separate denoise kernel.
struct DenoiseSumArgs {
    float *p[WT_MAX_SUM_PLANES];
    double tau[WT_MAX_SUM_PLANES];   // <= 0: significance identically one
    float wgt[WT_MAX_SUM_PLANES];
    int n, n_den, soft, write_back;
};

__global__ __launch_bounds__(256) void wt_denoise_sum_kernel(DenoiseSumArgs a, const float *noise,
                                                             float *out, int64_t n4)
{
#pragma clang fp contract(off)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        float4 nz = make_float4(1.f, 1.f, 1.f, 1.f);
        if (noise) nz = reinterpret_cast<const float4 *>(noise)[i];
        const float nn[4] = {nz.x, nz.y, nz.z, nz.w};
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < a.n; ++k) {
            const float4 v = wt_ldnt4(a.p[k] + 4 * i);
            float c[4] = {v.x, v.y, v.z, v.w};
            if (k < a.n_den) {
                const double tau = a.tau[k];
                const float tauf = (float)tau;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float sgn = tau > 0.0 ? wt_sig(c[j], tauf * nn[j], tau * (double)nn[j], a.soft) : 1.f;
                    c[j] = c[j] * (a.wgt[k] * sgn);
                }
                if (a.write_back) wt_stnt4(a.p[k] + 4 * i, make_float4(c[0], c[1], c[2], c[3]));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = k == 0 ? c[j] : acc[j] + c[j];
        }
        wt_stnt4(out + 4 * i, make_float4(acc[0], acc[1], acc[2], acc[3]));
    }
}

// K3/K4  significance / denoise.  mode 0: dst = sig ; mode 1: dst = c * (wgt*sig)
__global__ __launch_bounds__(256) void wt_signif_kernel(const float *c, const float *noise,
                                                        float *dst, int64_t n4, double tau,
                                                        float wgt, int soft, int mode)
{
    const float tauf = (float)tau;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4 *>(c)[i];
        float4 nz = make_float4(1.f, 1.f, 1.f, 1.f);
        if (noise) nz = reinterpret_cast<const float4 *>(noise)[i];
        const float in[4] = {v.x, v.y, v.z, v.w};
        const float nn[4] = {nz.x, nz.y, nz.z, nz.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float s = wt_sig(in[k], tauf * nn[k], tau * (double)nn[k], soft);
            o[k] = mode ? in[k] * (wgt * s) : s;
        }
        reinterpret_cast<float4 *>(dst)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// K6  wow per-scale update - watroo/utils.py:193-203 (see wt_wow_update in the header)
__global__ __launch_bounds__(256) void wt_wow_kernel(float *c, const float *power,
                                                     const float *noise, float *gamma,
                                                     int64_t n4, double tau, int soft,
                                                     float factor)
{
    const float tauf = (float)tau;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4 *>(c)[i];
        float4 nz = make_float4(1.f, 1.f, 1.f, 1.f), pw = nz, gm = make_float4(0, 0, 0, 0);
        if (noise) nz = reinterpret_cast<const float4 *>(noise)[i];
        if (power) pw = reinterpret_cast<const float4 *>(power)[i];
        if (gamma) gm = reinterpret_cast<const float4 *>(gamma)[i];
        float in[4] = {v.x, v.y, v.z, v.w};
        const float nn[4] = {nz.x, nz.y, nz.z, nz.w};
        const float pp[4] = {pw.x, pw.y, pw.z, pw.w};
        float gg[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            in[k] = wt_wow_point(in[k], pp[k], power != nullptr, nn[k], tau, tauf, soft, factor, gg[k]);
        reinterpret_cast<float4 *>(c)[i] = make_float4(in[0], in[1], in[2], in[3]);
        if (gamma) reinterpret_cast<float4 *>(gamma)[i] = make_float4(gg[0], gg[1], gg[2], gg[3]);
    }
}

// K8  gamma blend - watroo/utils.py:212-217
__global__ __launch_bounds__(256) void wt_gamma_kernel(float *recon, float *gamma, int64_t n4,
                                                       float gmin, float range, float inv_gamma,
                                                       float h)
{
#pragma clang fp contract(off)
    const float omh = 1.f - h;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float4 r = reinterpret_cast<const float4 *>(recon)[i];
        const float4 gq = reinterpret_cast<const float4 *>(gamma)[i];
        const float rr[4] = {r.x, r.y, r.z, r.w};
        float gg[4] = {gq.x, gq.y, gq.z, gq.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float t = (gg[k] - gmin) / range;
            t = t < 0.f ? 0.f : t;
            t = t > 1.f ? 1.f : t;
            t = powf(t, inv_gamma);
            gg[k] = t;
            o[k] = omh * rr[k] + h * t;
        }
        reinterpret_cast<float4 *>(recon)[i] = make_float4(o[0], o[1], o[2], o[3]);
        reinterpret_cast<float4 *>(gamma)[i] = make_float4(gg[0], gg[1], gg[2], gg[3]);
    }
}

// K11  generalized Anscombe - watroo/wavelets.py:14-21.  Host precomputes the scalar terms:
// forward: c1 = 3 alpha^2/8, c2 = sigma^2, c3 = alpha g ; inverse: c1 = alpha g, c2 = sigma^2,
// c3 = 3 alpha / 8.  Contraction is off so each numpy op rounds exactly as on the host.
__global__ __launch_bounds__(256) void wt_anscombe_kernel(const float *src, float *dst,
                                                          int64_t n4, float alpha, float c1,
                                                          float c2, float c3, int inverse)
{
#pragma clang fp contract(off)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4 *>(src)[i];
        const float in[4] = {v.x, v.y, v.z, v.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (inverse) {
                float t = alpha * in[k];
                t = t / 2.f;
                t = t * t;
                t = t + c1;
                t = t - c2;
                t = t - c3;
                o[k] = t / alpha;
            } else {
                float t = alpha * in[k];
                t = t + c1;
                t = t + c2;
                t = t - c3;
                t = t <= 0.f ? 0.f : t;
                o[k] = (2.f * sqrtf(t)) / alpha;
            }
        }
        reinterpret_cast<float4 *>(dst)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---------------------------------------------------------------------------------------------
// Richardson-Lucy support (watroo/utils.py:222-290)
// ---------------------------------------------------------------------------------------------
// Bilateral filtering of (Z, Y, X) cubes (atrous_convolution with the 3-D kernel,
// watroo/wavelets.py:74-105 called from :438-440): every tap of the K^3 dilated neighbourhood is
// range-weighted, out = (k_c I + sum k_t e_t I_t) / (k_c + sum k_t e_t), e_t = exp(-(I - I_t)^2 /
// (2 var)).  One voxel per thread, taps through L1/L2; cubes are small next to the 2-D images
// the tuned kernels serve, and the cost is the K^3 transcendental evaluations either way.
template <int K>
__global__ __launch_bounds__(256) void wt_bilateral3d_kernel(const float *in, const float *var, float *out,
                                                             int X, int P, int Y, int Z, int d, int border)
{
    constexpr int hw = K / 2;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= X) return;
    for (int row = blockIdx.y; row < Z * Y; row += gridDim.y) {
        const int z = row / Y, y = row - z * Y;
        const int64_t o = (int64_t)row * P + x;
        const float I = in[o];
        const float m = -0.5f / var[o];
        float num = wt_tap<K>(hw) * wt_tap<K>(hw) * wt_tap<K>(hw) * I;
        float den = wt_tap<K>(hw) * wt_tap<K>(hw) * wt_tap<K>(hw);
#pragma unroll 1
        for (int i = 0; i < K; ++i) {
            const int zz = wt_refl_b(z + (i - hw) * d, Z, d, border);
#pragma unroll 1
            for (int j = 0; j < K; ++j) {
                const int yy = wt_refl_b(y + (j - hw) * d, Y, d, border);
                const float kzy = wt_tap<K>(i) * wt_tap<K>(j);
                const float *r = in + ((int64_t)zz * Y + yy) * P;
#pragma unroll
                for (int l = 0; l < K; ++l) {
                    if (i == hw && j == hw && l == hw) continue;
                    const float It = r[wt_refl_b(x + (l - hw) * d, X, d, border)];
                    const float dl = I - It;
                    const float w = kzy * wt_tap<K>(l) * __expf(dl * dl * m);
                    num = fmaf(w, It, num);
                    den += w;
                }
            }
        }
        out[o] = num / den;
    }
}

// filter along axis 0 (axis == 0) or axis 1 (axis == 1, inside every slice) of a (Z, Y, X) cube
// stored as a (Z*Y) x X image, run-time taps: the second half of the per-slice 2-D filter and the
// third pass of convolution()'s 3-D branch (watroo/wavelets.py:46-63) for user-defined scaling
// functions
__global__ __launch_bounds__(256) void wt_custom_axis_kernel(const float *in, float *out, int W, int P, int Y,
                                                             int Z, int d, int border, CustomTaps t, int axis)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const int hw = t.n / 2;
    for (int row = blockIdx.y; row < Z * Y; row += gridDim.y) {
        const int z = row / Y, y = row - z * Y;
        float acc = 0.f;
        for (int j = 0; j < t.n; ++j) {
            const int zz = axis == 0 ? wt_refl_b(z + (j - hw) * d, Z, d, border) : z;
            const int yy = axis == 1 ? wt_refl_b(y + (j - hw) * d, Y, d, border) : y;
            const float v = in[((int64_t)zz * Y + yy) * P + x];
            acc = j == 0 ? t.k[0] * v : fmaf(t.k[j], v, acc);
        }
        out[(int64_t)row * P + x] = acc;
    }
}

// cv2.filter2D(src, -1, kernel, dst, (-1,-1), 0, BORDER_REFLECT) with an arbitrary small PSF
// (watroo/utils.py:257,286): correlation, anchor = kernel centre (k/2), symmetric border.
// 64 x 16 output tile + halo staged in LDS; the PSF taps are wave-uniform scalar loads.
#define WT_F2D_TW 64
#define WT_F2D_TH 16
// WRAP: periodic border (the circular convolution of the reference's rFFT path,
// watroo/utils.py:245-254,284), whole-image plans only; (ay, ax) = anchor of the correlation.
__device__ __forceinline__ int wt_wrap(int i, int n)
{
    const int m = i % n;
    return m < 0 ? m + n : m;
}

// (round 3) psf_pitch / ACCUM: a PSF beyond 4096 taps (or beyond the LDS tile) is applied in bands of
// rows and columns - each launch takes a kh x kw window of the full PSF (row pitch psf_pitch) with the
// anchor shifted into the window's frame (it may then lie outside the window) and adds to `out`
template <bool WRAP, bool ACCUM>
__global__ __launch_bounds__(256) void wt_filter2d_kernel(const float *in, float *out, Geo g,
                                                          const float *psf, int psf_pitch, int kh, int kw, int ay, int ax)
{
    extern __shared__ float tile[];
    const int tw = WT_F2D_TW + kw - 1, th = WT_F2D_TH + kh - 1;
    const int x0 = blockIdx.x * WT_F2D_TW, ly0 = blockIdx.y * WT_F2D_TH;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    for (int i = tid; i < tw * th; i += 256) {
        const int ty = i / tw, tx = i - ty * tw;
        if (WRAP) {
            const float *row = in + (int64_t)wt_wrap(ly0 + ty - ay, g.H) * g.P;
            tile[i] = row[wt_wrap(x0 + tx - ax, g.W)];
        } else {
            const float *row = wt_row(in, g, g.row0 + ly0 + ty - ay);
            tile[i] = row[wt_refl(x0 + tx - ax, g.W)];
        }
    }
    __syncthreads();
    const int x = x0 + threadIdx.x;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < kh; ++i)
        for (int j = 0; j < kw; ++j) {
            const float k = psf[i * psf_pitch + j];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[r] = fmaf(k, tile[(threadIdx.y * 4 + r + i) * tw + threadIdx.x + j], acc[r]);
        }
    if (x < g.W) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ly = ly0 + threadIdx.y * 4 + r;
            if (ly < g.nrows) out[(int64_t)ly * g.P + x] = ACCUM ? out[(int64_t)ly * g.P + x] + acc[r] : acc[r];
        }
    }
}

// elementwise binary ops of the RL iteration (watroo/utils.py:259,280-281,288)
enum { WT_OP_SUB = 0, WT_OP_ADD = 1, WT_OP_MUL = 2, WT_OP_DIV = 3, WT_OP_ADD_DIV = 4 };
__global__ __launch_bounds__(256) void wt_binary_kernel(const float *a, const float *b, float *dst,
                                                        int64_t n4, int op)
{
#pragma clang fp contract(off)
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float4 u = reinterpret_cast<const float4 *>(a)[i];
        const float4 v = reinterpret_cast<const float4 *>(b)[i];
        const float x[4] = {u.x, u.y, u.z, u.w}, y[4] = {v.x, v.y, v.z, v.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            switch (op) {
                case WT_OP_SUB: o[k] = x[k] - y[k]; break;
                case WT_OP_ADD: o[k] = x[k] + y[k]; break;
                case WT_OP_MUL: o[k] = x[k] * y[k]; break;
                case WT_OP_DIV: o[k] = x[k] / y[k]; break;
                default: o[k] = (x[k] + y[k]) / y[k]; break;   // res += phi; res /= phi
            }
        }
        reinterpret_cast<float4 *>(dst)[i] = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// multiresolution-support update of one residual plane (watroo/utils.py:263-276):
//   sig = significance(c);  hard: mrs = persistent ? max(mrs, sig) : sig ;  c *= mrs
//                           soft: mrs = persistent ? mrs * sig   : sig ;  c *= mrs ** inv_pow
__global__ __launch_bounds__(256) void wt_mrs_kernel(float *c, float *mrs, const float *noise,
                                                     int64_t n4, double tau, int soft,
                                                     int persistent, float inv_pow)
{
    const float tauf = (float)tau;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4 *>(c)[i];
        const float4 m4 = reinterpret_cast<const float4 *>(mrs)[i];
        float4 nz = make_float4(1.f, 1.f, 1.f, 1.f);
        if (noise) nz = reinterpret_cast<const float4 *>(noise)[i];
        float cc[4] = {v.x, v.y, v.z, v.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w};
        const float nn[4] = {nz.x, nz.y, nz.z, nz.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float sg = tau > 0.0 ? wt_sig(cc[k], tauf * nn[k], tau * (double)nn[k], soft) : 1.f;
            if (soft) {
                mm[k] = persistent ? mm[k] * sg : sg;
                cc[k] = cc[k] * powf(mm[k], inv_pow);
            } else {
                mm[k] = persistent ? fmaxf(mm[k], sg) : sg;
                cc[k] = cc[k] * mm[k];
            }
        }
        reinterpret_cast<float4 *>(c)[i] = make_float4(cc[0], cc[1], cc[2], cc[3]);
        reinterpret_cast<float4 *>(mrs)[i] = make_float4(mm[0], mm[1], mm[2], mm[3]);
    }
}

// 3-D branch of convolution() (watroo/wavelets.py:46-64): after the per-slice 2-D filter, a K-tap
// dilated filter along axis 0 (cv2.filter2D of every (Z, Y) slice with the (K', 1) kernel,
// BORDER_REFLECT).  The cube is stored as a (Z*Y) x X image, so axis 0 is rows Y apart.
template <int K>
__global__ __launch_bounds__(256) void wt_zfilter_kernel(const float *in, float *out, int64_t n4,
                                                         int P4, int Y, int Z, int d, int border)
{
    constexpr int hw = K / 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / P4), c4 = (int)(i % P4);
        const int z = row / Y, y = row - z * Y;
        float4 acc;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int zz = wt_refl_b(z + (j - hw) * d, Z, d, border);
            const float4 v = reinterpret_cast<const float4 *>(in)[((int64_t)zz * Y + y) * P4 + c4];
            acc = (j == 0) ? f4_scale(wt_tap<K>(0), v) : f4_fma(wt_tap<K>(j), v, acc);
        }
        reinterpret_cast<float4 *>(out)[i] = acc;
    }
}

// After a histogram pass: find the bin that holds rank k, fold it into the prefix, clear the bins.
// One block: every thread sums its run of bins, a block-wide scan of the 256 partial sums finds the
// one thread whose run holds the rank, and that thread walks its (<= 8) bins.
__global__ __launch_bounds__(256) void wt_select_step_kernel(uint32_t *hist, WtSelectState *st, int nbins, int shift, int last)
{
    __shared__ unsigned long long part[256];
    const int per = (nbins + 255) / 256;                 // <= WT_HIST_BINS / 256 = 8
    const int b0 = threadIdx.x * per;
    uint32_t h[WT_HIST_BINS / 256];
    unsigned long long s = 0;
#pragma unroll
    for (int i = 0; i < WT_HIST_BINS / 256; ++i) {
        h[i] = (i < per && b0 + i < nbins) ? hist[b0 + i] : 0u;
        s += h[i];
    }
    const unsigned long long k = st->k, cum_le = st->cum_le;
    const uint32_t prefix = st->prefix;
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {            // inclusive scan
        const unsigned long long v = threadIdx.x >= off ? part[threadIdx.x - off] : 0ull;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    const unsigned long long incl = part[threadIdx.x];
    unsigned long long cum = incl - s;
    if (k >= cum && k < incl) {                          // exactly one thread
#pragma unroll
        for (int i = 0; i < WT_HIST_BINS / 256; ++i) {
            if (k < cum + h[i]) {
                st->k = k - cum;
                st->cum_le = cum_le + cum + (last ? h[i] : 0);
                st->prefix = prefix | ((uint32_t)(b0 + i) << shift);
                break;
            }
            cum += h[i];
        }
    }
    if (threadIdx.x == 255 && k >= incl && st->failed == 0) st->failed = 1;  // rank beyond the population (NaN input);
                                                                              // (an earlier verdict - 3: window missed - stands)
    for (int i = threadIdx.x; i < nbins; i += 256) hist[i] = 0;     // ready for the next pass
}

#ifndef WT_HIST_UNROLL
#define WT_HIST_UNROLL 4      // 16-byte loads per thread and item; two items in flight (double-buffered)
#endif
// One level of the radix select: histogram of (|x| >> shift) & bin_mask over the elements whose bits
// under prefix_mask equal the selection prefix.
//  * REP interleaved copies of the LDS histogram (lh[bin * REP + (lane & (REP - 1))]).  The first
//    level bins EVERY element, and the magnitudes of a detail plane crowd into a few dozen bins (a
//    handful of exponents x 8 mantissa sub-bins): with one copy, same-address LDS atomics of a wave
//    serialise (round 2 counters: 47 % of the LDS cycles were bank conflicts).  Four copies put
//    neighbouring lanes on different addresses AND different banks (hot neighbouring bins times
//    four copies cover all 32 banks).  Later levels bin a few per cent of the elements: one copy.
//  * work items are (row, chunk of 256 * UNROLL float4) pairs; the loads of the NEXT item are
//    issued before the atomics of the current one (8 loads of 16 B in flight per thread instead of
//    4 with a full drain per iteration: the read stream was latency-bound at 4.7 TB/s).
//  * WIN (round 4, the select of a plane no fused pass has histogrammed): every element is binned into
//    the WINDOW of 21-bit keys that starts at *wbase (bins as in the riding histogram of wt_fused_kernel:
//    0 = below, 1 .. 2046 = key - base, 2047 = above; wt_select_window_step_kernel reads them) - the
//    first TWO levels of the select in one pass over the plane.
template <int REP, bool WIN = false>
__global__ __launch_bounds__(256) void wt_hist_kernel(const float *p, int nrows, int P4, int W,
                                                      uint32_t prefix_mask, const WtSelectState *st,
                                                      int shift, uint32_t bin_mask,
                                                      uint32_t *hist, const uint32_t *wbase = nullptr)
{
    const uint32_t prefix_val = WIN ? 0u : st->prefix & prefix_mask;      // wave-uniform scalar load
    const int win_lo = WIN ? (int)*wbase : 0;
    __shared__ uint32_t lh[WT_HIST_BINS * REP];
    for (int i = threadIdx.x; i < WT_HIST_BINS * REP; i += 256) lh[i] = 0;
    __syncthreads();
    constexpr int U = WT_HIST_UNROLL;
    const int X4 = (W + 3) >> 2;
    const int nchunk = (X4 + 256 * U - 1) / (256 * U);
    const int64_t nitems = (int64_t)nrows * nchunk;
    const int rep = threadIdx.x & (REP - 1);
    auto load = [&](int64_t item, float4 (&v)[U]) {
        const int r = (int)(item / nchunk), c = (int)(item - (int64_t)r * nchunk);
        const float *row = p + (int64_t)r * P4 * 4;
#pragma unroll
#ifdef WT_HIST_PLAIN_LOADS
        for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4 *>(row + 4 * min(c * 256 * U + 256 * u + (int)threadIdx.x, X4 - 1));
#else
        for (int u = 0; u < U; ++u) v[u] = wt_ldnt4(row + 4 * min(c * 256 * U + 256 * u + (int)threadIdx.x, X4 - 1));
#endif
    };
    auto bin = [&](int64_t item, const float4 (&v)[U]) {
        const int c = (int)(item % nchunk);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int xx = c * 256 * U + 256 * u + (int)threadIdx.x;
            const int nv = xx < X4 ? min(4, W - xx * 4) : 0;
            const uint32_t b[4] = {__float_as_uint(v[u].x), __float_as_uint(v[u].y),
                                   __float_as_uint(v[u].z), __float_as_uint(v[u].w)};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t w = b[k] & 0x7fffffffu;
                if constexpr (WIN) {
                    if (k < nv) atomicAdd(&lh[min(max((int)(w >> 10) - win_lo, 0), WT_HIST_BINS - 1) * REP + rep], 1u);
                } else {
                    if (k < nv && (w & prefix_mask) == prefix_val)
                        atomicAdd(&lh[((w >> shift) & bin_mask) * REP + rep], 1u);
                }
            }
        }
    };
    float4 va[U], vb[U];
    int64_t item = blockIdx.x;
    if (item < nitems) load(item, va);
    while (item < nitems) {                                  // two items per trip: no register copies
        const int64_t i1 = item + gridDim.x, i2 = i1 + gridDim.x;
        if (i1 < nitems) load(i1, vb);
        bin(item, va);
        if (i1 >= nitems) break;
        if (i2 < nitems) load(i2, va);
        bin(i1, vb);
        item = i2;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < WT_HIST_BINS; i += 256) {
        uint32_t n = 0;
#pragma unroll
        for (int k = 0; k < REP; ++k) n += lh[i * REP + k];
        if (n) atomicAdd(&hist[i], n);
    }
}

// smallest |x| bit pattern strictly greater than `than` (for the upper median when N is even);
// one global atomic per block
__global__ __launch_bounds__(256) void wt_min_greater_kernel(const float *p, int nrows, int P4,
                                                             int W, uint32_t than,
                                                             uint32_t *result)
{
    uint32_t best = 0xffffffffu;
    for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
        const float *row = p + (int64_t)r * P4 * 4;
        for (int x4 = threadIdx.x; x4 * 4 < W; x4 += 256) {
            const float4 v = wt_ldnt4(row + 4 * x4);
            const int nv = min(4, W - x4 * 4);
            const uint32_t b[4] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z),
                                   __float_as_uint(v.w)};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t u = b[k] & 0x7fffffffu;
                if (k < nv && u > than) best = min(best, u);
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) best = min(best, (uint32_t)__shfl_down((int)best, off));
    __shared__ uint32_t wb[4];
    if ((threadIdx.x & 63) == 0) wb[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        best = min(min(wb[0], wb[1]), min(wb[2], wb[3]));
        if (best != 0xffffffffu) atomicMin(result, best);
    }
}

// K7  {sum, sumsq, min, max}: fp64 sums, deterministic two-stage reduction (per-block partials
// over whole rows, then one block folds them in a fixed order).  Rows are walked with 2-D
// indices (no 64-bit modulo per element); min/max are taken in fp32, which is exact.
// Round 4: four 16-byte loads in flight per thread feeding four independent accumulator sets (folded
// in a fixed order at the end), nontemporal loads, 8 blocks per CU - the one-load loop with its
// dependent fp64 chains kept 16 KB in flight per CU and streamed at 0.52 of the HBM rate.
__global__ __launch_bounds__(256) void wt_reduce_kernel(const float *p, int nrows, int P4, int W,
                                                        double *partials)
{
    constexpr int U = 4;
    double sa[U] = {0.0, 0.0, 0.0, 0.0}, sb[U] = {0.0, 0.0, 0.0, 0.0};
    float mn = INFINITY, mx = -INFINITY;
    const int X4 = (W + 3) >> 2;
    // work items are (row, chunk of 256 * U float4) pairs dealt round-robin to the blocks (a fixed
    // assignment: deterministic sums); the loads of the NEXT item are issued before the current one is
    // folded - 8 loads of 16 B in flight per thread, as in the select passes
    const int nchunk = (X4 + 256 * U - 1) / (256 * U);
    const int64_t nitems = (int64_t)nrows * nchunk;
    auto load = [&](int64_t item, float4 (&v)[U]) {
        const int r = (int)(item / nchunk), c = (int)(item - (int64_t)r * nchunk);
        const float *row = p + (int64_t)r * P4 * 4;
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = wt_ldnt4(row + 4 * min(c * 256 * U + 256 * u + (int)threadIdx.x, X4 - 1));
    };
    auto fold = [&](int64_t item, const float4 (&v)[U]) {
        const int c = (int)(item % nchunk);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int xx = c * 256 * U + 256 * u + (int)threadIdx.x;
            const int nv = xx < X4 ? min(4, W - xx * 4) : 0;
            const float b[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < nv) {
                    const double t = (double)b[k];
                    sa[u] += t;
                    sb[u] = fma(t, t, sb[u]);
                    mn = fminf(mn, b[k]);
                    mx = fmaxf(mx, b[k]);
                }
        }
    };
    float4 va[U], vb[U];
    int64_t item = blockIdx.x;
    if (item < nitems) load(item, va);
    while (item < nitems) {
        const int64_t i1 = item + gridDim.x, i2 = i1 + gridDim.x;
        if (i1 < nitems) load(i1, vb);
        fold(item, va);
        if (i1 >= nitems) break;
        if (i2 < nitems) load(i2, va);
        fold(i1, vb);
        item = i2;
    }
    double s = (sa[0] + sa[1]) + (sa[2] + sa[3]), s2 = (sb[0] + sb[1]) + (sb[2] + sb[3]);
    __shared__ double red[4][2];
    __shared__ float redf[4][2];
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off);
        s2 += __shfl_down(s2, off);
        mn = fminf(mn, __shfl_down(mn, off));
        mx = fmaxf(mx, __shfl_down(mx, off));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[wave][0] = s; red[wave][1] = s2; redf[wave][0] = mn; redf[wave][1] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            s += red[w][0]; s2 += red[w][1];
            mn = fminf(mn, redf[w][0]); mx = fmaxf(mx, redf[w][1]);
        }
        double *o = partials + (int64_t)blockIdx.x * 4;
        o[0] = s; o[1] = s2; o[2] = (double)mn; o[3] = (double)mx;
    }
}

