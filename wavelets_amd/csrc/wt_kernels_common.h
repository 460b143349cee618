// Kernels and device helpers that more than one translation unit of libwatroo_hip.so launches (round 5: the
// host side is four units - wt_core / wt_transform / wt_apps / wt_f64): element widening, the local variance
// from two moments, the run-time-tap row filter and bilateral operator, the border rule of the tap-list
// operator and the operator itself, the sampling / window kernels of the median select and the final fold of
// the reductions.  The few that are not templates are `static`: every unit that includes this header carries
// its own copy.  gfx950 only; reference semantics are cited as file:line under /root/reference.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "wt_internal.h"
#include "wt_device.h"
#include "wt_stencil.h"

// ---------------------------------------------------------------------------------------------
// images of another element type, widened (and byte-swapped) on their way into a plane: wt_upload_int /
// wt64_upload_int.  One element per thread from a tightly packed staging copy of the host rows.
// SWAP: the elements are in the other byte order (FITS data is big-endian: astropy hands out '>i2', '>i4',
// '>f4', '>f8' arrays, all of which the reference recasts to float64, ref:297) - swapped here, per element.
template <int N> struct WtUintOf;
template <> struct WtUintOf<1> { typedef uint8_t T; };
template <> struct WtUintOf<2> { typedef uint16_t T; };
template <> struct WtUintOf<4> { typedef uint32_t T; };
template <> struct WtUintOf<8> { typedef uint64_t T; };
__device__ __forceinline__ uint8_t wt_bswap(uint8_t v) { return v; }
__device__ __forceinline__ uint16_t wt_bswap(uint16_t v) { return __builtin_bswap16(v); }
__device__ __forceinline__ uint32_t wt_bswap(uint32_t v) { return __builtin_bswap32(v); }
__device__ __forceinline__ uint64_t wt_bswap(uint64_t v) { return __builtin_bswap64(v); }

template <typename I, typename O, bool SWAP>
__global__ __launch_bounds__(256) void wt_from_elems_kernel(const I *src, O *dst, int W, int P, int nrows)
{
    typedef typename WtUintOf<sizeof(I)>::T U;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    for (int y = blockIdx.y; y < nrows; y += gridDim.y) {
        U raw = reinterpret_cast<const U *>(src)[(int64_t)y * W + x];
        if (SWAP) raw = wt_bswap(raw);
        dst[(int64_t)y * P + x] = (O)__builtin_bit_cast(I, raw);
    }
}

// variance plane from the two smoothed moments (sdev_loc, watroo/wavelets.py:24-32, with the
// factors of :434-436): dst = max(meansq - mean^2 -> 1e-20 if <= 0) * f1 * f2
static __global__ __launch_bounds__(256) void wt_var_moments_kernel(const float *mean, const float *meansq, float *dst,
                                                             int64_t n4, float f1, float f2, int take_sqrt = 0)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4 *>(mean)[i];
        const float4 b = reinterpret_cast<const float4 *>(meansq)[i];
        const float m[4] = {a.x, a.y, a.z, a.w}, q[4] = {b.x, b.y, b.z, b.w};
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = wt_var_point(q[k], m[k], f1, f2, take_sqrt);
        reinterpret_cast<float4 *>(dst)[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// User-defined scaling functions (AbstractScalingFunction subclasses, watroo/wavelets.py:152-229):
// the separable dilated filter with run-time taps, one pixel per thread, two passes (rows into a
// scratch plane, then columns + epilogue).  Generic and simple on purpose - the tuned kernels
// above are specialised to the two built-in families.
struct CustomTaps {
    float k[WT_MAX_CUSTOM_TAPS];
    int n;
};
// the plan's run-time taps as a kernel argument
static inline CustomTaps plan_taps(const wt_plan *p)
{
    CustomTaps t{};
    t.n = p->ntaps;
    for (int i = 0; i < p->ntaps; ++i) t.k[i] = p->taps[i];
    return t;
}

static __global__ __launch_bounds__(256) void wt_custom_rows_kernel(const float *in, float *tmp, Geo g, int d,
                                                             CustomTaps t, int square)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= g.W) return;
    const int hw = t.n / 2;
    for (int y = blockIdx.y; y < g.nrows; y += gridDim.y) {
        const float *row = in + (int64_t)y * g.P;
        float acc = 0.f;
        for (int j = 0; j < t.n; ++j) {
            float v = row[wt_refl_b(x + (j - hw) * d, g.W, d, g.border)];
            if (square) v *= v;
            acc = j == 0 ? t.k[0] * v : fmaf(t.k[j], v, acc);
        }
        tmp[(int64_t)y * g.P + x] = acc;
    }
}

// atrous_convolution(image, kernel, bilateral_variance, s) with run-time taps
// (watroo/wavelets.py:74-105): K^2 taps on an image (Z == 0; rows [g.row0, g.row0 + g.nrows) of
// it) or K^3 on a (Z, Y, X) cube.  The reference's tap loop is a TRUE CONVOLUTION - kernel index
// i pairs with the sample at offset (hw - i) * d (:87-91) - while the plan stores the taps in
// cv2.filter2D's correlation order, so tap i is t.k[i] here; `rev` = the plan's taps are stored
// reversed (plans of 1-D signals, whose smoothing is scipy's convolution).  Taps are visited in
// the reference's order (row-major kernel index).  One sample per thread.
static __global__ __launch_bounds__(256) void wt_bilateral_custom_kernel(const float *in, const float *var, float *out_c,
                                                                  float *out_w, Geo g, int Y, int Z, int d,
                                                                  CustomTaps t, int rev)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= g.W) return;
    const int n = t.n, hw = n / 2;
    const bool cube = Z > 0;
    const int nrows = cube ? Z * Y : g.nrows;
    const int H = cube ? Y : g.H;
    const float kc = t.k[hw];
    for (int row = blockIdx.y; row < nrows; row += gridDim.y) {
        const int z = cube ? row / Y : 0;
        const int y = cube ? row - z * Y : g.row0 + row;
        const int64_t o = (int64_t)row * g.P + x;
        const float I = in[o];
        const float m = -0.5f / var[o];
        float den = cube ? kc * kc * kc : kc * kc;
        float num = den * I;
        for (int iz = 0; iz < (cube ? n : 1); ++iz) {
            const int zz = cube ? wt_refl_b(z + (hw - iz) * d, Z, d, g.border) : 0;
            const float kz = cube ? t.k[rev ? n - 1 - iz : iz] : 1.f;
            for (int iy = 0; iy < n; ++iy) {
                const int yy = wt_refl_b(y + (hw - iy) * d, H, d, g.border);
                const float kzy = kz * t.k[rev ? n - 1 - iy : iy];
                const float *r = in + (cube ? ((int64_t)zz * Y + yy) : (int64_t)(yy - g.row0)) * g.P;
                for (int ix = 0; ix < n; ++ix) {
                    if (ix == hw && iy == hw && (!cube || iz == hw)) continue;
                    const float It = r[wt_refl_b(x + (hw - ix) * d, g.W, d, g.border)];
                    const float dl = I - It;
                    const float w = kzy * t.k[rev ? n - 1 - ix : ix] * __expf(dl * dl * m);
                    num = fmaf(w, It, num);
                    den += w;
                }
            }
        }
        const float c = num / den;
        if (out_w) out_w[o] = I - c;
        out_c[o] = c;
    }
}

// ---------------------------------------------------------------------------------------------
// Generic tap-list operator (round 3): the reference's atrous_convolution for ANY kernel and
// np.pad mode (watroo/wavelets.py:74-105) - non-separable kernels, even or large tap counts,
// 'symmetric' / 'reflect' / 'edge' / 'wrap' / 'constant' borders, with or without the range
// weights, on signals (1 x N), images and (Z, Y, X) cubes stored as (Z*Y) x X images.
//   plain:      out = kc * I + sum_t w_t * I_t                        (ref:79, 92-93; tap order kept)
//   bilateral:  out = (kc * I + sum_t e_t * I_t) / (kc + sum_t e_t),  e_t = w_t * exp(-(I - I_t)^2 / var / 2)
// I_t = the sample at offset (dz, dy, dx) under the border rule applied per axis.  One sample per
// thread, taps from a device list: the fallback for everything the tuned kernels do not cover
// (they take the separable built-in / user-defined taps under the symmetric border).
// ---------------------------------------------------------------------------------------------

// d: the dilation of the polyphase modes (the border rules of atrous_recursive, whose sub-arrays of
// stride d are each extended on their own: WT_PAD_POLY_SYMMETRIC / WT_PAD_POLY_MIRROR = wt_refl_b's
// border rules 1 / 3); unused by the np.pad modes
__device__ __forceinline__ int wt_pad_index(int i, int n, int mode, int d = 1)
{
    if ((unsigned)i < (unsigned)n) return i;
    switch (mode) {
        case WT_PAD_POLY_SYMMETRIC: return wt_refl_b(i, n, d, 1);
        case WT_PAD_POLY_MIRROR: return wt_refl_b(i, n, d, 3);
        case WT_PAD_SYMMETRIC: return wt_refl(i, n);
        case WT_PAD_REFLECT: {                           // d c b | a b c d | c b a  (no edge duplication)
            if (n == 1) return 0;
            const int p = 2 * n - 2;
            int m = i % p;
            if (m < 0) m += p;
            return m < n ? m : p - m;
        }
        case WT_PAD_EDGE: return i < 0 ? 0 : n - 1;
        case WT_PAD_WRAP: {
            int m = i % n;
            return m < 0 ? m + n : m;
        }
        default: return -1;                              // constant: the caller substitutes the fill value
    }
}

template <typename T>
__global__ __launch_bounds__(256) void wt_taps_kernel(const T *in, const T *var, T *out, int W, int P, int Y, int Z,
                                                      const int32_t *offs, const T *wts, int ntaps, T kc,
                                                      int has_center, int mode, T cval, int dil)
{
#pragma clang fp contract(off)
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    for (int row = blockIdx.y; row < Z * Y; row += gridDim.y) {
        const int z = row / Y, y = row - z * Y;
        const T I = in[(int64_t)row * P + x];
        T acc = has_center ? kc * I : (T)0, norm = has_center ? kc : (T)0;
        const T iv = var ? var[(int64_t)row * P + x] : (T)1;
        for (int t = 0; t < ntaps; ++t) {
            const int zz = wt_pad_index(z + offs[3 * t], Z, mode, dil);
            const int yy = wt_pad_index(y + offs[3 * t + 1], Y, mode, dil);
            const int xx = wt_pad_index(x + offs[3 * t + 2], W, mode, dil);
            const T v = (zz < 0 || yy < 0 || xx < 0) ? cval : in[((int64_t)zz * Y + yy) * P + xx];
            if (var) {
                const T dlt = I - v;
                const T e = wts[t] * exp(-(dlt * dlt) / iv / (T)2);     // ref:97
                norm = norm + e;
                acc = acc + v * e;
            } else {
                acc = acc + v * wts[t];                                  // ref:93
            }
        }
        out[(int64_t)row * P + x] = var ? acc / norm : acc;
    }
}

// ---------------------------------------------------------------------------------------------
// K2  exact median of |x| by radix select on the fp32 bit pattern (non-negative floats order
// like their uint32 bits) - np.median(np.abs(data[0])), watroo/wavelets.py:127.
// One histogram pass per digit (11 + 10 + 10 bits); LDS-privatised bins, one global atomic per
// non-empty bin per block.  Pixels in the pitch padding (x >= W) are masked.
// ---------------------------------------------------------------------------------------------
// Selection state kept on the device between the passes of the radix select (wt_abs_median): the
// passes chain on the stream without a host round trip; the host reads the state once at the end.
struct WtSelectState {
    unsigned long long k;        // rank still to find among the elements matching `prefix`
    unsigned long long cum_le;   // elements below the selected bins so far (+ the last bin's population at the end)
    uint32_t prefix;             // bits fixed so far
    uint32_t failed;             // rank not found (NaN input)
};

// Round 4: where the 2048 bins of the riding histogram should sit.  With the top 11 magnitude bits as
// the key (8 exponent + 3 mantissa bits) a detail plane populates a few dozen bins and two more passes
// over the plane must resolve the other 20 bits.  This kernel - ONE workgroup, before the transform -
// computes |w_0| = |I - h * I| at 4096 pixels of a regular grid straight from the input image (scale 0
// needs a K x K neighbourhood; wt_median_sample_kernel), takes the median of that sample
// (wt_median_window_kernel, one workgroup) and centres a window of 2046 keys of 21
// bits (relative resolution 1.2e-4, +-12 % around the estimate) on it: *base = first key of the window.
// The standard error of a 4096-sample median is ~2 % of sigma, so the true median lies inside the
// window except for pathological images - which the select detects (rank in bin 0 / 2047) and redoes
// with its ordinary three passes.  The prediction only places bins; it never enters a result.
// window key of a sample: the top 21 magnitude bits of a float, the top 22 of a double
__device__ __forceinline__ uint32_t wt_window_key(float v) { return (__float_as_uint(v) & 0x7fffffffu) >> 10; }
__device__ __forceinline__ uint32_t wt_window_key(double v)
{
    return (uint32_t)(((unsigned long long)__double_as_longlong(v) & 0x7fffffffffffffffull) >> 41);
}

template <int K, typename T>
__global__ __launch_bounds__(64) void wt_median_sample_kernel(const T *in, Geo g, uint32_t *keys)
{
    // one sample per thread, 64 workgroups of one wave: the K * K loads of a sample are independent and
    // the 4096 samples spread over the chip (as ONE workgroup this step took 0.08 ms - more than the
    // pass over the plane it saves)
    constexpr int hw = K / 2;
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int sy = i >> 6, sx = i & 63;
    const int y = (int)(((int64_t)(2 * sy + 1) * g.H) >> 7), x = (int)(((int64_t)(2 * sx + 1) * g.W) >> 7);
    T v[K][K];
#pragma unroll
    for (int a = 0; a < K; ++a) {
        const T *row = in + (int64_t)(wt_refl(y + a - hw, g.H) - g.row0) * g.P;
#pragma unroll
        for (int b = 0; b < K; ++b) v[a][b] = row[wt_refl(x + b - hw, g.W)];
    }
    T acc = (T)0;
#pragma unroll
    for (int a = 0; a < K; ++a) {
        T r = (T)0;
#pragma unroll
        for (int b = 0; b < K; ++b) r = fma((T)wt_tap<K>(b), v[a][b], r);
        acc = fma((T)wt_tap<K>(a), r, acc);
    }
    keys[i] = wt_window_key(v[hw][hw] - acc);
}

// the same sample taken from a PLANE that already holds the coefficients (select without a riding
// histogram: bilateral / recursive / generic transforms, Coefficients built from arrays)
template <typename T>
__global__ __launch_bounds__(64) void wt_plane_sample_kernel(const T *plane, int nrows, int W, int P, uint32_t *keys)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    const int sy = i >> 6, sx = i & 63;
    const int y = (int)(((int64_t)(2 * sy + 1) * nrows) >> 7), x = (int)(((int64_t)(2 * sx + 1) * W) >> 7);
    keys[i] = wt_window_key(plane[(int64_t)y * P + x]);
}

// median of the 4096 window keys (<= 22 bits: two levels of 11) -> *base = first key of the window
static __global__ __launch_bounds__(1024) void wt_median_window_kernel(const uint32_t *keys, uint32_t *base)
{
    constexpr int NS = 4096;
    __shared__ uint32_t key[NS];
    __shared__ uint32_t lh[WT_HIST_BINS];
    __shared__ uint32_t part[1024];
    __shared__ uint32_t sh_k, sh_prefix;
    for (int i = threadIdx.x; i < NS; i += 1024) key[i] = keys[i];
    uint32_t k = NS / 2 - 1, prefix = 0, known = 0;
    for (int lvl = 0; lvl < 2; ++lvl) {
        const int shift = lvl == 0 ? 11 : 0;
        const uint32_t mask = 0x7ffu;
        for (int i = threadIdx.x; i < WT_HIST_BINS; i += 1024) lh[i] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < NS; i += 1024)
            if ((key[i] & known) == prefix) atomicAdd(&lh[(key[i] >> shift) & mask], 1u);
        __syncthreads();
        const uint32_t h0 = lh[2 * threadIdx.x], h1 = lh[2 * threadIdx.x + 1];
        part[threadIdx.x] = h0 + h1;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const uint32_t v = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
            __syncthreads();
            part[threadIdx.x] += v;
            __syncthreads();
        }
        const uint32_t incl = part[threadIdx.x], excl = incl - h0 - h1;
        if (k >= excl && k < incl) {
            const bool second = k >= excl + h0;
            sh_k = k - excl - (second ? h0 : 0);
            sh_prefix = prefix | ((uint32_t)(2 * threadIdx.x + (second ? 1 : 0)) << shift);
        }
        __syncthreads();
        k = sh_k;
        prefix = sh_prefix;
        known |= mask << shift;
        __syncthreads();
    }
    if (threadIdx.x == 0) *base = (uint32_t)max((int)prefix - WT_HIST_BINS / 2, 0);
}

// Step after a WINDOWED riding histogram (bins: 0 = below the window, 1 .. 2046 = the 21-bit keys base + bin,
// 2047 = above): the bin that holds rank k fixes the top 21 bits at once; a rank in bin 0 / 2047 means the
// prediction missed (failed = 3: the host redoes the select with its ordinary passes).  Clears the bins.
static __global__ __launch_bounds__(256) void wt_select_window_step_kernel(uint32_t *hist, WtSelectState *st, const uint32_t *base)
{
    __shared__ unsigned long long part[256];
    constexpr int per = WT_HIST_BINS / 256;
    const int b0 = threadIdx.x * per;
    uint32_t h[per];
    unsigned long long s = 0;
#pragma unroll
    for (int i = 0; i < per; ++i) {
        h[i] = hist[b0 + i];
        s += h[i];
    }
    const unsigned long long k = st->k;
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const unsigned long long v = threadIdx.x >= off ? part[threadIdx.x - off] : 0ull;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    const unsigned long long incl = part[threadIdx.x];
    unsigned long long cum = incl - s;
    if (k >= cum && k < incl) {
#pragma unroll
        for (int i = 0; i < per; ++i) {
            if (k < cum + h[i]) {
                const int bin = b0 + i;
                if (bin == 0 || bin == WT_HIST_BINS - 1) {
                    st->failed = 3;
                } else {
                    st->k = k - cum;
                    st->cum_le = cum;
                    st->prefix = (*base + (uint32_t)bin) << 10;
                }
                break;
            }
            cum += h[i];
        }
    }
    if (threadIdx.x == 255 && k >= incl) st->failed = 1;
    for (int i = threadIdx.x; i < WT_HIST_BINS; i += 256) hist[i] = 0;
}

// one 256-thread block: thread t folds partials t, t+256, ... in index order, then a fixed tree
static __global__ __launch_bounds__(256) void wt_reduce_final_kernel(const double *partials, int nblocks, double *out)
{
    double s = 0.0, s2 = 0.0, mn = INFINITY, mx = -INFINITY;
    for (int b = threadIdx.x; b < nblocks; b += 256) {
        s += partials[b * 4 + 0];
        s2 += partials[b * 4 + 1];
        mn = fmin(mn, partials[b * 4 + 2]);
        mx = fmax(mx, partials[b * 4 + 3]);
    }
    __shared__ double red[256][4];
    red[threadIdx.x][0] = s; red[threadIdx.x][1] = s2; red[threadIdx.x][2] = mn; red[threadIdx.x][3] = mx;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            red[threadIdx.x][0] += red[threadIdx.x + off][0];
            red[threadIdx.x][1] += red[threadIdx.x + off][1];
            red[threadIdx.x][2] = fmin(red[threadIdx.x][2], red[threadIdx.x + off][2]);
            red[threadIdx.x][3] = fmax(red[threadIdx.x][3], red[threadIdx.x + off][3]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = red[0][0]; out[1] = red[0][1]; out[2] = red[0][2]; out[3] = red[0][3];
    }
}
