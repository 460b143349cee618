// Generic kernels of the float64 engine (wt_f64.hip): one sample (or one 16-byte pair) per thread, run-time taps -
// the dilated row / column / axis filters and the marching form, the tap-for-tap bilateral operator (images and
// cubes), the pointwise operators (binary, thresholds, thresholds + sum, wow update, gamma blend, fill, Anscombe,
// variance from moments), the reductions, the small-PSF correlation and the support update of richardson_lucy.
// Images with a built-in family run on the tuned kernels instead (wt_fused.h, wt_stencil.h, wt_bilateral64.h);
// these serve signals, cubes, user-defined taps and the A/B switches.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include "wt_internal.h"
#include "wt_device.h"
#include "wt_math64.h"
#include "wt_kernels_common.h"

typedef double wt_ntd2 __attribute__((ext_vector_type(2)));     // streaming 16-byte accesses

struct Taps64 {
    double k[WT64_MAX_TAPS];
    int n;
};

// ---------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------
// dilated filter along x: tmp[y][x] = sum_j k_j * in[y][R(x + (j - hw) d)]   (correlation order, like
// cv2.filter2D, watroo/wavelets.py:39-45); square: filter in^2 (sdev_loc, :26)
__global__ __launch_bounds__(256) void wt64_rows_kernel(const double *in, double *tmp, Geo g, int d, Taps64 t, int square)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= g.W) return;
    const int hw = t.n / 2;
    for (int y = blockIdx.y; y < g.nrows; y += gridDim.y) {
        const double *row = in + (int64_t)y * g.P;
        double acc = 0.0;
        for (int j = 0; j < t.n; ++j) {
            double v = row[wt_refl_b(x + (j - hw) * d, g.W, d, g.border)];
            if (square) v *= v;
            acc = j == 0 ? t.k[0] * v : fma(t.k[j], v, acc);
        }
        tmp[(int64_t)y * g.P + x] = acc;
    }
}

// The same row filter with TWO pixels per thread (round 4, late): for an even dilation and an even width
// the taps of the pixel pair (x, x + 1), x even, are the aligned pairs (x + (j - hw) d, + 1) - one 16-byte
// load each instead of two 8-byte ones through separate reflections.  The one-pixel kernels run 1.9 TB/s
// (vector-memory issue, not HBM); same FMA chains per pixel: identical bits.
__global__ __launch_bounds__(256) void wt64_rows2_kernel(const double *in, double *tmp, Geo g, int d, Taps64 t, int square)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (x >= g.W) return;
    const int hw = t.n / 2;
    for (int y = blockIdx.y; y < g.nrows; y += gridDim.y) {
        const double *row = in + (int64_t)y * g.P;
        double2 acc = make_double2(0.0, 0.0);
        for (int j = 0; j < t.n; ++j) {
            const int xo = x + (j - hw) * d;
            double2 v;
            if (xo >= 0 && xo + 1 < g.W) v = *reinterpret_cast<const double2 *>(row + xo);
            else v = make_double2(row[wt_refl_b(xo, g.W, d, g.border)], row[wt_refl_b(xo + 1, g.W, d, g.border)]);
            if (square) v = make_double2(v.x * v.x, v.y * v.y);
            acc = j == 0 ? make_double2(t.k[0] * v.x, t.k[0] * v.y) : make_double2(fma(t.k[j], v.x, acc.x), fma(t.k[j], v.y, acc.y));
        }
        *reinterpret_cast<double2 *>(tmp + (int64_t)y * g.P + x) = acc;
    }
}

// ... and the column filter of an image (axis 1 of a one-slice cube), two pixels per thread
__global__ __launch_bounds__(256) void wt64_cols2_kernel(const double *in, double *out, const double *cen, double *out_w, int W, int P,
                                                         int nrows, int d, int border, Taps64 t)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (x >= W) return;
    const int hw = t.n / 2;
    for (int y = blockIdx.y; y < nrows; y += gridDim.y) {
        double2 acc = make_double2(0.0, 0.0);
        for (int j = 0; j < t.n; ++j) {
            const double2 v = *reinterpret_cast<const double2 *>(in + (int64_t)wt_refl_b(y + (j - hw) * d, nrows, d, border) * P + x);
            acc = j == 0 ? make_double2(t.k[0] * v.x, t.k[0] * v.y) : make_double2(fma(t.k[j], v.x, acc.x), fma(t.k[j], v.y, acc.y));
        }
        const int64_t o = (int64_t)y * P + x;
        if (out_w) {
            const double2 c = *reinterpret_cast<const double2 *>(cen + o);
            *reinterpret_cast<double2 *>(out_w + o) = make_double2(c.x - acc.x, c.y - acc.y);
        }
        *reinterpret_cast<double2 *>(out + o) = acc;
    }
}


// dilated filter along axis 1 (inside every slice: axis == 1) or axis 0 (across slices) of a
// (Z, Y, X) cube stored as a (Z*Y) x X image; an image is the cube with Z = 1
// (watroo/wavelets.py:35-63).  out_w != nullptr: also the detail plane cen - result (:442).
__global__ __launch_bounds__(256) void wt64_axis_kernel(const double *in, double *out, const double *cen, double *out_w,
                                                        int W, int P, int Y, int Z, int d, int border, Taps64 t, int axis)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const int hw = t.n / 2;
    for (int row = blockIdx.y; row < Z * Y; row += gridDim.y) {
        const int z = row / Y, y = row - z * Y;
        double acc = 0.0;
        for (int j = 0; j < t.n; ++j) {
            const int zz = axis == 0 ? wt_refl_b(z + (j - hw) * d, Z, d, border) : z;
            const int yy = axis == 1 ? wt_refl_b(y + (j - hw) * d, Y, d, border) : y;
            const double v = in[((int64_t)zz * Y + yy) * P + x];
            acc = j == 0 ? t.k[0] * v : fma(t.k[j], v, acc);
        }
        const int64_t o = (int64_t)row * P + x;
        if (out_w) out_w[o] = cen[o] - acc;
        out[o] = acc;
    }
}

// One scale of an image in ONE kernel: a thread owns a column x and one chunk of one polyphase row
// chain y = q, q + d, q + 2d, ... and marches down it.  Every step it filters the next row of the
// chain along x (K taps through L1 / L2), pushes the result into a K-deep register window and
// emits the vertical filter of the window: every input row is row-filtered once per chain instead
// of K times, nothing goes through a scratch plane (24 B per sample of HBM traffic instead of two
// passes of 5 loads + 1 store).  Same arithmetic and order as wt64_rows_kernel + wt64_axis_kernel
// (rows first, then columns, FMA chains in tap order): identical results.
__global__ __launch_bounds__(256) void wt64_chain_kernel(const double *in, double *out_c, double *out_w, Geo g, int d, Taps64 t,
                                                         int S, int chunks)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= g.W) return;
    const int n = t.n, hw = n / 2;
    for (int item = blockIdx.y; item < d * chunks; item += gridDim.y) {
        const int q = item % d, c = item / d;
        if (q >= g.H) continue;
        const int n_q = (g.H - q + d - 1) / d;           // chain length
        const int r0 = c * S, r1 = min(r0 + S, n_q);
        if (r0 >= r1) continue;
        int xi[WT64_MAX_TAPS];
#pragma unroll
        for (int j = 0; j < WT64_MAX_TAPS; ++j) xi[j] = j < n ? wt_refl_b(x + (j - hw) * d, g.W, d, g.border) : 0;
        double win[WT64_MAX_TAPS];
#pragma unroll
        for (int j = 0; j < WT64_MAX_TAPS; ++j) win[j] = 0.0;
        // the taps of the NEXT row are loaded before this row is consumed (one row in flight)
        double nx[WT64_MAX_TAPS];
        {
            const double *row = in + (int64_t)wt_refl_b(q + d * (r0 - hw), g.H, d, g.border) * g.P;
#pragma unroll
            for (int j = 0; j < WT64_MAX_TAPS; ++j) nx[j] = j < n ? row[xi[j]] : 0.0;
        }
        for (int tt = r0 - hw; tt < r1 + hw; ++tt) {
            double cu[WT64_MAX_TAPS];
#pragma unroll
            for (int j = 0; j < WT64_MAX_TAPS; ++j) cu[j] = nx[j];
            {
                const double *row = in + (int64_t)wt_refl_b(q + d * min(tt + 1, r1 + hw - 1), g.H, d, g.border) * g.P;
#pragma unroll
                for (int j = 0; j < WT64_MAX_TAPS; ++j)
                    if (j < n) nx[j] = row[xi[j]];
            }
            double h = 0.0;
#pragma unroll
            for (int j = 0; j < WT64_MAX_TAPS; ++j)
                if (j < n) h = j == 0 ? t.k[0] * cu[0] : fma(t.k[j], cu[j], h);
            // window: win[0] oldest ... win[n-1] newest
#pragma unroll
            for (int j = 0; j < WT64_MAX_TAPS - 1; ++j)
                if (j < n - 1) win[j] = win[j + 1];
#pragma unroll
            for (int j = 0; j < WT64_MAX_TAPS; ++j)
                if (j == n - 1) win[j] = h;
            const int r = tt - hw;                       // chain element whose window is complete
            if (r >= r0) {
                double v = 0.0;
#pragma unroll
                for (int j = 0; j < WT64_MAX_TAPS; ++j)
                    if (j < n) v = j == 0 ? t.k[0] * win[0] : fma(t.k[j], win[j], v);
                const int64_t o = (int64_t)(q + d * r) * g.P + x;
                if (out_w) out_w[o] = in[o] - v;
                out_c[o] = v;
            }
        }
    }
}

// atrous_convolution(image, kernel, bilateral_variance, s) in float64 (watroo/wavelets.py:74-105):
// K^2 taps on an image (Z == 0) or K^3 on a (Z, Y, X) cube, range-weighted.  The reference's tap loop
// is a true convolution (kernel index i pairs with the sample at offset (hw - i) * d, :87-91) while
// the plan stores the taps in correlation order; rev = the plan's taps are stored reversed (1-D
// signals).  Taps in the reference's order.  One sample per thread.
__global__ __launch_bounds__(256) void wt64_bilateral_kernel(const double *in, const double *var, double *out, Geo g, int Y, int Z, int d,
                                                             Taps64 t, int rev)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= g.W) return;
    const int n = t.n, hw = n / 2;
    const bool cube = Z > 0;
    const int nrows = cube ? Z * Y : g.nrows;
    const int H = cube ? Y : g.H;
    const double kc = t.k[hw];
    for (int row = blockIdx.y; row < nrows; row += gridDim.y) {
        const int z = cube ? row / Y : 0;
        const int y = cube ? row - z * Y : row;
        const int64_t o = (int64_t)row * g.P + x;
        const double I = in[o];
        const double m = -0.5 / var[o];
        double den = cube ? kc * kc * kc : kc * kc;
        double num = den * I;
        for (int iz = 0; iz < (cube ? n : 1); ++iz) {
            const int zz = cube ? wt_refl_b(z + (hw - iz) * d, Z, d, g.border) : 0;
            const double kz = cube ? t.k[rev ? n - 1 - iz : iz] : 1.0;
            for (int iy = 0; iy < n; ++iy) {
                const int yy = wt_refl_b(y + (hw - iy) * d, H, d, g.border);
                const double kzy = kz * t.k[rev ? n - 1 - iy : iy];
                const double *r = in + (cube ? ((int64_t)zz * Y + yy) : (int64_t)yy) * g.P;
                for (int ix = 0; ix < n; ++ix) {
                    if (ix == hw && iy == hw && (!cube || iz == hw)) continue;
                    const double It = r[wt_refl_b(x + (hw - ix) * d, g.W, d, g.border)];
                    const double dl = I - It;
                    const double w = kzy * t.k[rev ? n - 1 - ix : ix] * exp(dl * dl * m);
                    num = fma(w, It, num);
                    den += w;
                }
            }
        }
        out[o] = num / den;
    }
}

// pointwise: 0 add, 1 sub, 2 mul, 3 div, 4 (a + b) / b (watroo/utils.py:280-281)
__global__ __launch_bounds__(256) void wt64_binary_kernel(const double *a, const double *b, double *dst, int W, int P, int nrows, int op)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    for (int y = blockIdx.y; y < nrows; y += gridDim.y) {
        const int64_t o = (int64_t)y * P + x;
        const double u = a[o], v = b[o];
        dst[o] = op == 0 ? u + v : op == 1 ? u - v : op == 2 ? u * v : op == 3 ? u / v : (u + v) / v;
    }
}

// (wt_erf64, wt_sig64, wt_sig64_inv: wt_math64.h)

// Coefficients.significance / denoise (watroo/wavelets.py:129-149): mode 0: dst = significance;
// mode 1: dst = c * (wgt * significance).  tau <= 0: significance one.  noise: optional per-pixel map
// that multiplies tau (:133).  A lane owns two samples (16-byte accesses; planes are contiguous with an
// even pitch, n2 = double2 groups of the plane) - with wt_erf64 the kernel is a memory stream.
__global__ __launch_bounds__(256) void wt64_signif_kernel(const double *c, const double *noise, double *dst, int64_t n2, double tau,
                                                          double wgt, int soft, int mode)
{
    const double inv_tau = 1.0 / tau;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
        const double2 v = reinterpret_cast<const double2 *>(c)[i];
        double2 sg = make_double2(1.0, 1.0);
        if (tau > 0.0) {
            if (noise) {
                const double2 nz = reinterpret_cast<const double2 *>(noise)[i];
                sg = make_double2(wt_sig64(v.x, tau * nz.x, soft), wt_sig64(v.y, tau * nz.y, soft));
            } else {
                sg = make_double2(wt_sig64_inv(v.x, tau, inv_tau, soft), wt_sig64_inv(v.y, tau, inv_tau, soft));
            }
        }
        reinterpret_cast<double2 *>(dst)[i] = mode ? make_double2(v.x * (wgt * sg.x), v.y * (wgt * sg.y)) : sg;
    }
}

// Coefficients.denoise over the first n_den planes fused with np.sum(planes, axis=0) (wt_denoise_sum in
// float64): plane k < n_den becomes c * (wgt_k * significance_k) - the expression of wt64_signif_kernel,
// identical bits - and is written back if asked; the sum runs in plane order.
struct DenoiseSum64Args {
    double *p[32];
    double tau[32], inv_tau[32], wgt[32];      // inv_tau = 1 / tau (IEEE division on the host, as the kernels' own)
    int n, n_den, soft, write_back;
};
__global__ __launch_bounds__(256) void wt64_denoise_sum_kernel(DenoiseSum64Args a, const double *noise, double *dst, int64_t n2)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
        double2 nz = make_double2(1.0, 1.0);
        if (noise) nz = reinterpret_cast<const double2 *>(noise)[i];
        double2 acc = make_double2(0.0, 0.0);
        for (int k = 0; k < a.n; ++k) {
            const wt_ntd2 raw = __builtin_nontemporal_load(reinterpret_cast<const wt_ntd2 *>(a.p[k]) + i);   // (read exactly once)
            double2 v = make_double2(raw.x, raw.y);
            if (k < a.n_den) {
                double2 sg = make_double2(1.0, 1.0);
                if (a.tau[k] > 0.0) {
                    if (noise) sg = make_double2(wt_sig64(v.x, a.tau[k] * nz.x, a.soft), wt_sig64(v.y, a.tau[k] * nz.y, a.soft));
                    else sg = make_double2(wt_sig64_inv(v.x, a.tau[k], a.inv_tau[k], a.soft), wt_sig64_inv(v.y, a.tau[k], a.inv_tau[k], a.soft));
                }
                v = make_double2(v.x * (a.wgt[k] * sg.x), v.y * (a.wgt[k] * sg.y));
                if (a.write_back) reinterpret_cast<double2 *>(a.p[k])[i] = v;
            }
            acc = k == 0 ? v : make_double2(acc.x + v.x, acc.y + v.y);
        }
        __builtin_nontemporal_store((wt_ntd2){acc.x, acc.y}, reinterpret_cast<wt_ntd2 *>(dst) + i);
    }
}

// wow per-scale update (watroo/utils.py:193-203): c <- c * significance; gamma += c;
// c <- c * factor / sqrt(clip(power, 1e-15)).  power / noise / gamma may be null.
// one coefficient of the update; `pw`: its local power (has_power) - shared by the pointwise kernel and the
// column pass that forms the power itself (wt64_wow_axis_kernel): identical bits
__device__ __forceinline__ double wt64_wow_point(double t, bool has_power, double pw, const double *noise, double *gamma, int64_t o,
                                                 double tau, int soft, double factor)
{
    if (tau > 0.0) {
        const double tt = noise ? tau * noise[o] : tau;
        t = t * wt_sig64(t, tt, soft);
    }
    if (gamma) gamma[o] = gamma[o] + t;
    double q = factor;
    if (has_power) {
        const double lp = pw <= 0.0 ? 1e-15 : pw;
        q = factor * wt_rsq64(lp);       // (as wt_wow_point<double> of wt_stencil.h: the fused update gives identical bits)
    }
    return t * q;
}

__global__ __launch_bounds__(256) void wt64_wow_kernel(double *c, const double *power, const double *noise, double *gamma, int W, int P,
                                                       int nrows, double tau, int soft, double factor)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    for (int y = blockIdx.y; y < nrows; y += gridDim.y) {
        const int64_t o = (int64_t)y * P + x;
        c[o] = wt64_wow_point(c[o], power != nullptr, power ? power[o] : 0.0, noise, gamma, o, tau, soft, factor);
    }
}

// The column pass of conv_s(c^2) with the update as its epilogue (wt64_wow_scale): `rows` holds the row-
// filtered squares (wt64_rows_kernel, square = 1); the local power of a pixel is formed in registers and
// the coefficient is updated IN PLACE (this pass reads neighbours from `rows` only) - no power plane, one
// pass over the coefficients less than smooth + wt64_wow_update.
__global__ __launch_bounds__(256) void wt64_wow_axis_kernel(const double *rows, double *c, const double *noise, double *gamma, int W, int P,
                                                            int nrows, int d, int border, Taps64 t, double tau, int soft, double factor)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const int hw = t.n / 2;
    for (int y = blockIdx.y; y < nrows; y += gridDim.y) {
        double acc = 0.0;
        for (int j = 0; j < t.n; ++j) {
            const double v = rows[(int64_t)wt_refl_b(y + (j - hw) * d, nrows, d, border) * P + x];
            acc = j == 0 ? t.k[0] * v : fma(t.k[j], v, acc);
        }
        const int64_t o = (int64_t)y * P + x;
        c[o] = wt64_wow_point(c[o], true, acc, noise, gamma, o, tau, soft, factor);
    }
}

// gamma blend (watroo/utils.py:212-217)
__global__ __launch_bounds__(256) void wt64_gamma_kernel(double *recon, const double *gamma, int W, int P, int nrows, double gmin,
                                                         double range, double inv_gamma, double h)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    for (int y = blockIdx.y; y < nrows; y += gridDim.y) {
        const int64_t o = (int64_t)y * P + x;
        double t = (gamma[o] - gmin) / range;
        t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
        t = pow(t, inv_gamma);
        recon[o] = (1.0 - h) * recon[o] + h * t;
    }
}

__global__ __launch_bounds__(256) void wt64_fill_kernel(double *dst, int W, int P, int nrows, double value)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    for (int y = blockIdx.y; y < nrows; y += gridDim.y) dst[(int64_t)y * P + x] = value;
}

// {sum, sumsq, min, max}: per-block partials over whole rows, folded by wt_reduce_final_kernel; four
// 16-byte loads in flight per thread feeding independent accumulators (folded in a fixed order)
__global__ __launch_bounds__(256) void wt64_reduce_kernel(const double *p, int nrows, int P, int W, double *partials)
{
    constexpr int U = 4;
    double sa[U] = {0.0, 0.0, 0.0, 0.0}, sb[U] = {0.0, 0.0, 0.0, 0.0}, mn = INFINITY, mx = -INFINITY;
    const int X2 = (W + 1) / 2;
    for (int r = blockIdx.x; r < nrows; r += gridDim.x) {
        const double *row = p + (int64_t)r * P;
        for (int x2 = threadIdx.x; x2 < X2; x2 += 256 * U) {
            double2 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const double2 *>(row + 2 * min(x2 + 256 * u, X2 - 1));
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int xx = x2 + 256 * u;
                const double e[2] = {v[u].x, v[u].y};
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    if (xx < X2 && 2 * xx + k < W) {
                        sa[u] += e[k];
                        sb[u] = fma(e[k], e[k], sb[u]);
                        mn = fmin(mn, e[k]);
                        mx = fmax(mx, e[k]);
                    }
            }
        }
    }
    double s = (sa[0] + sa[1]) + (sa[2] + sa[3]), s2 = (sb[0] + sb[1]) + (sb[2] + sb[3]);
    __shared__ double red[4][4];
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off);
        s2 += __shfl_down(s2, off);
        mn = fmin(mn, __shfl_down(mn, off));
        mx = fmax(mx, __shfl_down(mx, off));
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[wave][0] = s; red[wave][1] = s2; red[wave][2] = mn; red[wave][3] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { s += red[w][0]; s2 += red[w][1]; mn = fmin(mn, red[w][2]); mx = fmax(mx, red[w][3]); }
        double *o = partials + (int64_t)blockIdx.x * 4;
        o[0] = s; o[1] = s2; o[2] = mn; o[3] = mx;
    }
}

// cv2.filter2D(src, -1, kernel, dst, (-1,-1), 0, BORDER_REFLECT) with an arbitrary small kernel
// (watroo/utils.py:257,286), or - wrap - the periodic correlation that the reference's rFFT products
// are (:245-254, 284); (ay, ax) = anchor.  One sample per thread, taps from a device buffer.
__global__ __launch_bounds__(256) void wt64_filter2d_kernel(const double *in, double *out, Geo g, const double *psf, int kh, int kw, int ay,
                                                            int ax, int wrap)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= g.W) return;
    for (int y = blockIdx.y; y < g.nrows; y += gridDim.y) {
        double acc = 0.0;
        for (int i = 0; i < kh; ++i) {
            int yy = y + i - ay;
            if (wrap) { yy %= g.H; if (yy < 0) yy += g.H; } else yy = wt_refl(yy, g.H);
            const double *row = in + (int64_t)yy * g.P;
            for (int j = 0; j < kw; ++j) {
                int xx = x + j - ax;
                if (wrap) { xx %= g.W; if (xx < 0) xx += g.W; } else xx = wt_refl(xx, g.W);
                acc = fma(psf[i * kw + j], row[xx], acc);
            }
        }
        out[(int64_t)y * g.P + x] = acc;
    }
}

// multiresolution-support update of a residual plane (watroo/utils.py:263-276), as wt_mrs_kernel
__global__ __launch_bounds__(256) void wt64_mrs_kernel(double *c, double *mrs, const double *noise, int W, int P, int nrows, double tau,
                                                       int soft, int persistent, double inv_pow)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    for (int y = blockIdx.y; y < nrows; y += gridDim.y) {
        const int64_t o = (int64_t)y * P + x;
        const double v = c[o];
        double sg = 1.0;
        if (tau > 0.0) {
            const double tt = noise ? tau * noise[o] : tau;
            sg = wt_sig64(v, tt, soft);
        }
        double m = mrs[o];
        if (soft) {
            m = persistent ? m * sg : sg;
            c[o] = v * pow(m, inv_pow);
        } else {
            m = persistent ? fmax(m, sg) : sg;
            c[o] = v * m;
        }
        mrs[o] = m;
    }
}

struct Sum64Args {
    const double *p[32];
    int n;
};
// np.sum(planes, axis=0) in plane order (watroo/utils.py:98); a lane owns two samples (16-byte accesses)
// As wt_plane_sum_kernel (round 5): one 16-byte group per thread - a large grid of short-lived waves keeps the
// most loads in flight for this n-reads-1-write stream - and streaming (nontemporal) accesses: planes read
// exactly once should not displace cache lines (8192^2, 12 planes: 1.33 -> 1.1x ms).
__global__ __launch_bounds__(256) void wt64_plane_sum_kernel(Sum64Args a, double *dst, int64_t n2)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
        wt_ntd2 acc = __builtin_nontemporal_load(reinterpret_cast<const wt_ntd2 *>(a.p[0]) + i);
        for (int k = 1; k < a.n; ++k) {
            const wt_ntd2 v = __builtin_nontemporal_load(reinterpret_cast<const wt_ntd2 *>(a.p[k]) + i);
            acc = acc + v;
        }
        __builtin_nontemporal_store(acc, reinterpret_cast<wt_ntd2 *>(dst) + i);
    }
}

// generalized_anscombe (watroo/wavelets.py:14-21)
__global__ __launch_bounds__(256) void wt64_anscombe_kernel(const double *src, double *dst, int W, int P, int nrows, double alpha,
                                                            double g, double sigma, int inverse)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    for (int y = blockIdx.y; y < nrows; y += gridDim.y) {
        const int64_t o = (int64_t)y * P + x;
        const double v = src[o];
        double r;
        if (inverse) {
            const double h = alpha * v / 2.0;
            r = (h * h + alpha * g - sigma * sigma - 3.0 * alpha / 8.0) / alpha;
        } else {
            double dum = alpha * v + 3.0 * alpha * alpha / 8.0 + sigma * sigma - alpha * g;
            if (dum <= 0.0) dum = 0.0;
            r = 2.0 * sqrt(dum) / alpha;
        }
        dst[o] = r;
    }
}

// sdev_loc (watroo/wavelets.py:24-32) from the two smoothed moments
__global__ __launch_bounds__(256) void wt64_var_kernel(const double *mean, const double *meansq, double *dst, int W, int P, int nrows,
                                                       double f1, double f2, int take_sqrt)
{
#pragma clang fp contract(off)
    // (the reference multiplies, then subtracts, ref:27; and wt_var_point of wt_stencil.h must give the same bits)
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    for (int y = blockIdx.y; y < nrows; y += gridDim.y) {
        const int64_t o = (int64_t)y * P + x;
        double v = meansq[o] - mean[o] * mean[o];
        if (v <= 0.0) v = 1e-20;
        if (take_sqrt) v = sqrt(v);
        dst[o] = (v * f1) * f2;
    }
}

// wt64_wow_axis_kernel with two pixels per thread (even widths): 16-byte accesses throughout
__global__ __launch_bounds__(256) void wt64_wow_axis2_kernel(const double *rows, double *c, const double *noise, double *gamma, int W, int P,
                                                             int nrows, int d, int border, Taps64 t, double tau, int soft, double factor)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 2;
    if (x >= W) return;
    const int hw = t.n / 2;
    for (int y = blockIdx.y; y < nrows; y += gridDim.y) {
        double2 acc = make_double2(0.0, 0.0);
        for (int j = 0; j < t.n; ++j) {
            const double2 v = *reinterpret_cast<const double2 *>(rows + (int64_t)wt_refl_b(y + (j - hw) * d, nrows, d, border) * P + x);
            acc = j == 0 ? make_double2(t.k[0] * v.x, t.k[0] * v.y) : make_double2(fma(t.k[j], v.x, acc.x), fma(t.k[j], v.y, acc.y));
        }
        const int64_t o = (int64_t)y * P + x;
        const double2 cc = *reinterpret_cast<const double2 *>(c + o);
        double2 r;
        r.x = wt64_wow_point(cc.x, true, acc.x, noise, gamma, o, tau, soft, factor);
        r.y = wt64_wow_point(cc.y, true, acc.y, noise, gamma, o + 1, tau, soft, factor);
        *reinterpret_cast<double2 *>(c + o) = r;
    }
}

