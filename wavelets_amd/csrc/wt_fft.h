// Circular products through a hand-written FFT (round 4): richardson_lucy(fft=True) of the reference
// forms  irfft2(rfft2(psi) * fft_psf)  and  irfft2(rfft2(res) * conj(fft_psf))  (watroo/utils.py:245-254,
// 284).  Rounds 2-3 evaluated them as direct periodic correlations - exact, but O(kh * kw) per pixel; a
// 129 x 129 PSF costs 16 641 taps per pixel per product.  Here, for images whose sides have no prime factor
// above 5 (round 5: mixed radix 2 / 3 / 5; powers of two keep the radix-2 kernel of round 4), up to 8192 per side:
//   forward:  rows FFT (length W, one workgroup per row, in LDS)  ->  transpose  ->  rows FFT (H)
//   product:  fused into the load of the first inverse pass (spectrum * kernel spectrum, or its conjugate)
//   inverse:  rows IFFT (H)  ->  transpose  ->  rows IFFT (W), real part scaled by 1 / (H W) into the plane
// Full complex transforms of the real planes (the Hermitian half is not exploited: six memory-bound
// kernels over 8 (float) or 16 (double) bytes per pixel against a direct form that is compute-bound by
// three orders of magnitude more work; simplicity wins).  Twiddles exp(-2 pi i k / n) come from a table
// computed in double on the host.  Templated on the element type: wt_plan (float) and wt_plan64 (double).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>

#include "wt_internal.h"

template <typename T> struct WtCx;
template <> struct WtCx<float> { typedef float2 C; };
template <> struct WtCx<double> { typedef double2 C; };
__device__ __forceinline__ float2 wt_cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 wt_cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ float2 wt_cmake(float x, float y) { return make_float2(x, y); }
__device__ __forceinline__ double2 wt_cmake(double x, double y) { return make_double2(x, y); }

#define WT_FFT_MAX_N 8192
#define WT_FFT_IN_REAL 1      // read a real plane (pitch in elements), imaginary part 0
#define WT_FFT_OUT_REAL 2     // write the real part into a real plane
#define WT_FFT_MUL 4          // multiply the loaded element by mul[...] ...
#define WT_FFT_MUL_CONJ 8     // ... or by its conjugate

// One FFT of length n per workgroup (row `blockIdx.x`), radix-2 decimation in time in LDS: the row is
// loaded in bit-reversed order, log2(n) butterfly stages with one barrier each, then stored.  INV: the
// conjugate twiddles (the caller folds the 1 / n factors into `scale`).
template <typename T, bool INV>
__global__ __launch_bounds__(512) void wt_fft_rows_kernel(const void *in, void *out, int n, int log2n, int in_pitch, int out_pitch,
                                                          const typename WtCx<T>::C *tw, const typename WtCx<T>::C *mul, int flags, T scale)
{
    typedef typename WtCx<T>::C C;
    extern __shared__ unsigned char wt_fft_lds[];
    C *s = reinterpret_cast<C *>(wt_fft_lds);
    const int row = blockIdx.x, nt = blockDim.x;
    for (int i = threadIdx.x; i < n; i += nt) {
        C v;
        if (flags & WT_FFT_IN_REAL) {
            v.x = reinterpret_cast<const T *>(in)[(int64_t)row * in_pitch + i];
            v.y = (T)0;
        } else {
            v = reinterpret_cast<const C *>(in)[(int64_t)row * in_pitch + i];
        }
        if (flags & (WT_FFT_MUL | WT_FFT_MUL_CONJ)) {
            C m = mul[(int64_t)row * n + i];
            if (flags & WT_FFT_MUL_CONJ) m.y = -m.y;
            v = wt_cmul(v, m);
        }
        s[__brev((unsigned)i) >> (32 - log2n)] = v;
    }
    __syncthreads();
    for (int st = 0; st < log2n; ++st) {
        const int half = 1 << st, tstep = n >> (st + 1);
        for (int j = threadIdx.x; j < (n >> 1); j += nt) {
            const int pos = j & (half - 1), i0 = ((j >> st) << (st + 1)) + pos, i1 = i0 + half;
            C w = tw[pos * tstep];
            if (INV) w.y = -w.y;
            const C a = s[i0], t = wt_cmul(w, s[i1]);
            s[i0] = wt_cmake(a.x + t.x, a.y + t.y);
            s[i1] = wt_cmake(a.x - t.x, a.y - t.y);
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += nt) {
        const C v = s[i];
        if (flags & WT_FFT_OUT_REAL) reinterpret_cast<T *>(out)[(int64_t)row * out_pitch + i] = v.x * scale;
        else reinterpret_cast<C *>(out)[(int64_t)row * out_pitch + i] = wt_cmake(v.x * scale, v.y * scale);
    }
}

// The same for lengths n = r_0 r_1 ... r_{k-1} with r_t in {2, 3, 5} (round 5): decimation in time in place.
// Stage t combines r_t transforms of length L_t = r_0 ... r_{t-1}, stored side by side, into one of length
// L_t r_t:  X[k + q L_t] = sum_m w_r^(m q) (w_(L_t r_t)^(m k) Y_m[k]),  Y_m[k] at offset m L_t + k - the outputs
// take the places of the inputs.  The input permutation that makes every stage's operands contiguous blocks is
// the mixed-radix digit reversal: with i = m_{k-1} + r_{k-1} (m_{k-2} + r_{k-2} (...)) element i goes to
// sum_t m_t L_t (bit reversal when every r_t is 2).  Twiddles: tw[j] = exp(-2 pi i j / n), j < n.
struct WtFftFactors {
    int nf, n2;          // factors; how many of them (the first n2) are 2
    int r[16];
};

template <int R, typename C, bool INV>
__device__ __forceinline__ void wt_fft_butterfly(C *s, int base, int L, int k, int step, int n, const C *tw)
{
    C b[R];
#pragma unroll
    for (int m = 0; m < R; ++m) {
        const C a = s[base + m * L];
        if (m == 0) {
            b[m] = a;
        } else {
            C w = tw[m * k * step];
            if (INV) w.y = -w.y;
            b[m] = wt_cmul(w, a);
        }
    }
    if (R == 2) {
        s[base] = wt_cmake(b[0].x + b[1].x, b[0].y + b[1].y);
        s[base + L] = wt_cmake(b[0].x - b[1].x, b[0].y - b[1].y);
        return;
    }
    C o[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        C acc = b[0];
#pragma unroll
        for (int m = 1; m < R; ++m) {
            C w = tw[((m * q) % R) * (n / R)];
            if (INV) w.y = -w.y;
            const C t = wt_cmul(w, b[m]);
            acc = wt_cmake(acc.x + t.x, acc.y + t.y);
        }
        o[q] = acc;
    }
#pragma unroll
    for (int q = 0; q < R; ++q) s[base + q * L] = o[q];
}

template <typename T, bool INV>
__global__ __launch_bounds__(512) void wt_fft_rows_mixed_kernel(const void *in, void *out, int n, WtFftFactors fa, int in_pitch, int out_pitch,
                                                                const typename WtCx<T>::C *tw, const typename WtCx<T>::C *mul, int flags, T scale)
{
    typedef typename WtCx<T>::C C;
    extern __shared__ unsigned char wt_fft_lds[];
    C *s = reinterpret_cast<C *>(wt_fft_lds);
    const int row = blockIdx.x, nt = blockDim.x;
    for (int i = threadIdx.x; i < n; i += nt) {
        C v;
        if (flags & WT_FFT_IN_REAL) {
            v.x = reinterpret_cast<const T *>(in)[(int64_t)row * in_pitch + i];
            v.y = (T)0;
        } else {
            v = reinterpret_cast<const C *>(in)[(int64_t)row * in_pitch + i];
        }
        if (flags & (WT_FFT_MUL | WT_FFT_MUL_CONJ)) {
            C m = mul[(int64_t)row * n + i];
            if (flags & WT_FFT_MUL_CONJ) m.y = -m.y;
            v = wt_cmul(v, m);
        }
        // digit reversal: the digits of i from the last factor to the first, weighted by the block lengths L_t.
        // The factors come 2s first (wt_fft_factor): the 5s and 3s are peeled off by constant divisors, what is
        // left (< 2^n2) is the bit reversal of the radix-2 digits.
        int rem = i, p = 0, L = n;
        for (int t = fa.nf - 1; t >= fa.n2; --t) {
            int m;
            if (fa.r[t] == 3) { L /= 3; m = rem % 3; rem /= 3; }
            else { L /= 5; m = rem % 5; rem /= 5; }
            p += m * L;
        }
        if (fa.n2) p += (int)(__brev((unsigned)rem) >> (32 - fa.n2));
        s[p] = v;
    }
    __syncthreads();
    int L = 1;
    for (int t = 0; t < fa.nf; ++t) {
        const int r = fa.r[t], Ln = L * r, step = n / Ln;
        const bool pow2 = (L & (L - 1)) == 0;
        const int sh = __ffs(L) - 1;
        for (int j = threadIdx.x; j < n / r; j += nt) {
            const int g = pow2 ? j >> sh : j / L, k = j - g * L, base = g * Ln + k;
            if (r == 2) wt_fft_butterfly<2, C, INV>(s, base, L, k, step, n, tw);
            else if (r == 3) wt_fft_butterfly<3, C, INV>(s, base, L, k, step, n, tw);
            else wt_fft_butterfly<5, C, INV>(s, base, L, k, step, n, tw);
        }
        __syncthreads();
        L = Ln;
    }
    for (int i = threadIdx.x; i < n; i += nt) {
        const C v = s[i];
        if (flags & WT_FFT_OUT_REAL) reinterpret_cast<T *>(out)[(int64_t)row * out_pitch + i] = v.x * scale;
        else reinterpret_cast<C *>(out)[(int64_t)row * out_pitch + i] = wt_cmake(v.x * scale, v.y * scale);
    }
}

// out[c][r] = in[r][c] for a rows x cols complex array (32 x 32 tiles through LDS, padded against bank conflicts)
template <typename T>
__global__ __launch_bounds__(256) void wt_fft_transpose_kernel(const typename WtCx<T>::C *in, typename WtCx<T>::C *out, int rows, int cols)
{
    typedef typename WtCx<T>::C C;
    __shared__ C tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8 threads
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < rows && c0 + tx < cols) tile[k][tx] = in[(int64_t)(r0 + k) * cols + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < cols && r0 + tx < rows) out[(int64_t)(c0 + k) * rows + r0 + tx] = tile[tx][k];
}

// n = a product of 2s, 3s and 5s: the factors, 2s first (their stages - and, as long as only 2s came before, the
// next one - split the butterfly index by shifts; the input permutation of the radix-2 part is a bit reversal)
static inline bool wt_fft_factor(int n, WtFftFactors &f)
{
    f.nf = f.n2 = 0;
    if (n < 2 || n > WT_FFT_MAX_N) return false;
    for (int r : {2, 3, 5})
        while (n % r == 0) {
            if (f.nf == 16) return false;
            f.r[f.nf++] = r;
            if (r == 2) ++f.n2;
            n /= r;
        }
    return n == 1;
}
static inline bool wt_fft_size_ok(int H, int W)
{
    WtFftFactors f;
    return wt_fft_factor(H, f) && wt_fft_factor(W, f);
}
static inline int wt_ilog2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }

template <typename T>
static int wt_fft_prepare(wt_ctx *c, WtFftState &f, int H, int W, std::vector<void *> &owner)
{
    typedef typename WtCx<T>::C C;
    if (f.a && f.H == H && f.W == W) return 0;
    if (f.a) WT_FAIL("wt_fft: the plan's geometry changed");
    if (!wt_fft_size_ok(H, W)) WT_FAIL("wt_fft: a side of the %d x %d image has a prime factor above 5 (or lies outside 2 .. %d)", H, W, WT_FFT_MAX_N);
    WT_HIP(hipSetDevice(c->device));
    auto alloc = [&](void **p, size_t bytes) -> int {
        WT_HIP(hipMalloc(p, bytes));
        owner.push_back(*p);
        return 0;
    };
    const size_t nc = (size_t)H * W * sizeof(C);
    WT_TRY(alloc(&f.a, nc));
    WT_TRY(alloc(&f.b, nc));
    WT_TRY(alloc(&f.spec, nc));
    auto table = [&](void **p, int n) -> int {          // exp(-2 pi i k / n), k < n, in double on the host
        std::vector<C> t((size_t)n);
        for (int k = 0; k < n; ++k) {
            const double ang = -2.0 * M_PI * (double)k / (double)n;
            t[k].x = (T)std::cos(ang);
            t[k].y = (T)std::sin(ang);
        }
        WT_TRY(alloc(p, t.size() * sizeof(C)));
        WT_HIP(hipMemcpy(*p, t.data(), t.size() * sizeof(C), hipMemcpyHostToDevice));
        return 0;
    };
    WT_TRY(table(&f.tw_w, W));
    WT_TRY(table(&f.tw_h, H));
    f.H = H;
    f.W = W;
    return 0;
}

template <typename T, bool INV>
static int wt_fft_rows(wt_ctx *c, const void *in, void *out, int nrows, int n, int in_pitch, int out_pitch, const void *tw, const void *mul,
                       int flags, T scale)
{
    typedef typename WtCx<T>::C C;
    const size_t lds = (size_t)n * sizeof(C);
    const int threads = std::max(64, std::min(512, (n / 2 + 63) / 64 * 64));
    if ((n & (n - 1)) == 0) {                            // powers of two: the radix-2 kernel
        if (lds > 64 * 1024)
            WT_HIP(hipFuncSetAttribute((const void *)wt_fft_rows_kernel<T, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((wt_fft_rows_kernel<T, INV>), dim3(nrows), dim3(threads), lds, c->stream, in, out, n, wt_ilog2(n), in_pitch, out_pitch,
                           (const C *)tw, (const C *)mul, flags, scale);
    } else {
        WtFftFactors fa;
        if (!wt_fft_factor(n, fa)) WT_FAIL("wt_fft: length %d has a prime factor above 5", n);
        if (lds > 64 * 1024)
            WT_HIP(hipFuncSetAttribute((const void *)wt_fft_rows_mixed_kernel<T, INV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((wt_fft_rows_mixed_kernel<T, INV>), dim3(nrows), dim3(threads), lds, c->stream, in, out, n, fa, in_pitch, out_pitch,
                           (const C *)tw, (const C *)mul, flags, scale);
    }
    WT_HIP(hipGetLastError());
    return 0;
}

// spectrum (transposed, W x H) of the real H x W plane `src` (pitch P) -> dstC
template <typename T>
static int wt_fft_forward(wt_ctx *c, WtFftState &f, const T *src, int P, void *dstC)
{
    typedef typename WtCx<T>::C C;
    const int H = f.H, W = f.W;
    WT_TRY((wt_fft_rows<T, false>(c, src, f.a, H, W, P, W, f.tw_w, nullptr, WT_FFT_IN_REAL, (T)1)));
    hipLaunchKernelGGL(wt_fft_transpose_kernel<T>, dim3((W + 31) / 32, (H + 31) / 32), dim3(256), 0, c->stream, (const C *)f.a, (C *)f.b, H, W);
    WT_HIP(hipGetLastError());
    return wt_fft_rows<T, false>(c, f.b, dstC, W, H, H, H, f.tw_h, nullptr, 0, (T)1);
}

// The kernel spectrum of the plan <- FFT2 of the real plane `src` (the PSF placed periodically by the caller)
template <typename T>
static int wt_fft_set_spectrum(wt_ctx *c, WtFftState &f, const T *src, int P)
{
    ProfScope ps(c, "wt_fft_kernels");
    WT_TRY(wt_fft_forward<T>(c, f, src, P, f.spec));
    f.have_spec = true;
    return 0;
}

// dst = real(IFFT2(FFT2(src) * K))  (conj: * conj(K)), K the plan's kernel spectrum
template <typename T>
static int wt_fft_apply_t(wt_ctx *c, WtFftState &f, const T *src, T *dst, int P, int conj)
{
    typedef typename WtCx<T>::C C;
    if (!f.have_spec) WT_FAIL("wt_fft_apply: no kernel spectrum (wt_fft_spectrum first)");
    const int H = f.H, W = f.W;
    ProfScope ps(c, "wt_fft_kernels");
    WT_TRY(wt_fft_forward<T>(c, f, src, P, f.a));                     // a: W x H spectrum of src  (uses a, b, then a)
    // first inverse pass: rows of the transposed spectrum (length H), times the kernel spectrum
    WT_TRY((wt_fft_rows<T, true>(c, f.a, f.b, W, H, H, H, f.tw_h, f.spec, conj ? WT_FFT_MUL_CONJ : WT_FFT_MUL, (T)1)));
    hipLaunchKernelGGL(wt_fft_transpose_kernel<T>, dim3((H + 31) / 32, (W + 31) / 32), dim3(256), 0, c->stream, (const C *)f.b, (C *)f.a, W, H);
    WT_HIP(hipGetLastError());
    return wt_fft_rows<T, true>(c, f.a, dst, H, W, W, P, f.tw_w, nullptr, WT_FFT_OUT_REAL, (T)(1.0 / ((double)H * (double)W)));
}
