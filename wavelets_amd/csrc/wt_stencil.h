// Per-scale stencil kernels of the a-trous engine for gfx950: the chain march, the lattice kernel and the
// row kernel, written once for both element types (round 5).  A lane moves 16 bytes per access either
// way: T = float - a lane owns 4 adjacent pixels (float4) - or T = double - 2 pixels (double2); the
// register windows cost the same VGPRs, the float64 kernels are the float32 design at twice the bytes
// per pixel.  The reference computes float64 / integer inputs in float64 (watroo/wavelets.py:297,
// 319-320), so this is what every int16 / FITS frame runs on.
//
// Reference semantics restated by each kernel are cited as file:line under /root/reference.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "wt_internal.h"
#include "wt_device.h"
#include "wt_math64.h"

// Row pointer for GLOBAL row gy (any integer): reflect on the global image, then map into
// this strip's buffer (rows outside the strip live in the halo margins).
template <typename T>
__device__ __forceinline__ const T *wt_row(const T *base, const Geo &g, int gy)
{
    const int ry = wt_refl(gy, g.H);
    return base + (int64_t)(ry - g.row0) * g.P;
}

// border-mode aware forms used by the single-scale operators (d = dilation of the operator)
template <typename T>
__device__ __forceinline__ const T *wt_row_b(const T *base, const Geo &g, int gy, int d)
{
    const int ry = wt_refl_b(gy, g.H, d, g.border);
    return base + (int64_t)(ry - g.row0) * g.P;
}
// descriptor of one row (P elements of T) behind a wave-uniform pointer: range check = the row's pitch
template <typename T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wt_row_rsrc(const T *row, int P)
{
    const uint64_t ra = (uint64_t)row;
    const uint64_t ua = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(ra >> 32)) << 32) |
                        (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ra);
    return __builtin_amdgcn_make_buffer_rsrc((void *)ua, 0, P * (int)sizeof(T), 0x00020000);
}
// symmetric reflection of the bilateral kernels' two border rules (Geo::border 0 / 1; wt_refl_b without the
// 'mirror' rules the launch code refuses)
__device__ __forceinline__ int wt_refl_01(int i, int n, int d, int border)
{
    if ((unsigned)i < (unsigned)n) return i;
    if (border == 0) return wt_refl(i, n);
    int o = i % d;
    if (o < 0) o += d;
    return o + d * wt_refl((i - o) / d, (n - o + d - 1) / d);
}

__device__ __forceinline__ float4 wt_load4_b(const float *row, int xo, int W, int d, int border)
{
    if (xo >= 0 && xo + 3 < W) return *reinterpret_cast<const float4 *>(row + xo);
    return make_float4(row[wt_refl_b(xo, W, d, border)], row[wt_refl_b(xo + 1, W, d, border)],
                       row[wt_refl_b(xo + 2, W, d, border)], row[wt_refl_b(xo + 3, W, d, border)]);
}
// one lane's group of pixels starting at pixel xo (a multiple of the group size), reflected at the border
__device__ __forceinline__ float4 wt_loadv_b(const float *row, int xo, int W, int d, int border)
{
    return wt_load4_b(row, xo, W, d, border);
}
__device__ __forceinline__ double2 wt_loadv_b(const double *row, int xo, int W, int d, int border)
{
    if (xo >= 0 && xo + 1 < W) return *reinterpret_cast<const double2 *>(row + xo);
    return make_double2(row[wt_refl_b(xo, W, d, border)], row[wt_refl_b(xo + 1, W, d, border)]);
}

// 4 consecutive pixels starting at pixel xo (xo % 4 == 0) of a row, reflected at the image
// border.  Interior: one 16-byte load.
__device__ __forceinline__ float4 wt_load4(const float *row, int xo, int W)
{
    if (xo >= 0 && xo + 3 < W) return *reinterpret_cast<const float4 *>(row + xo);
    return make_float4(row[wt_refl(xo, W)], row[wt_refl(xo + 1, W)], row[wt_refl(xo + 2, W)],
                       row[wt_refl(xo + 3, W)]);
}

typedef unsigned int wt_su4 __attribute__((ext_vector_type(4)));
typedef float wt_sf4 __attribute__((ext_vector_type(4)));
typedef double wt_sd2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ wt_su4 wt_bits16(float4 v)
{
    const wt_sf4 t = {v.x, v.y, v.z, v.w};
    return __builtin_bit_cast(wt_su4, t);
}
__device__ __forceinline__ wt_su4 wt_bits16(double2 v)
{
    const wt_sd2 t = {v.x, v.y};
    return __builtin_bit_cast(wt_su4, t);
}
// 16-byte store of one lane's pixel group at pixel x of an image row (row = wave-uniform pointer to the
// row's first pixel, P = row pitch in elements) through a raw buffer descriptor: a lane that must not
// write (lane_ok false, or x beyond the row) gets an out-of-range offset and the hardware drops the
// store.  No exec-mask branch around the store, so the compiler's vmcnt bookkeeping stays exact:
// behind a branch it has to assume the store may not have been issued and every wait for the
// next row's loads also waits for this row's stores.  A group that straddles W writes into the
// row's pitch padding (allocated, never read as image data), like the fused passes.
// nt: streaming (nontemporal) store - the host sets it for planes far larger than the caches
// (wave-uniform), where write-once outputs only displace useful lines (wow 8192^2: -3 %)
template <typename T, typename V>
__device__ __forceinline__ void wt_storev(T *row, int x, int P, V v, int nt = 0, bool lane_ok = true)
{
    const uint64_t ra = (uint64_t)row;
    // (the builtin returns int: go through unsigned, or the low word sign-extends into the high one)
    const uint64_t ua = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(ra >> 32)) << 32) |
                        (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)ra);
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)ua, 0, P * (int)sizeof(T), 0x00020000);
    const unsigned voff = (lane_ok && x >= 0) ? (unsigned)x * (unsigned)sizeof(T) : 0xfffffff0u;
    if (nt) __builtin_amdgcn_raw_buffer_store_b128(wt_bits16(v), r, voff, 0, 2);
    else __builtin_amdgcn_raw_buffer_store_b128(wt_bits16(v), r, voff, 0, 0);
}
__device__ __forceinline__ void wt_store4(float *row, int x, int P, float4 v, int nt = 0, bool lane_ok = true)
{
    wt_storev<float, float4>(row, x, P, v, nt, lane_ok);
}

__device__ __forceinline__ float wt_sig(float c, float tau, double taud, int soft)
{
    // Coefficients.significance - watroo/wavelets.py:137-141
    if (soft) return erff(fabsf(c / tau));
    return ((double)fabsf(c) > taud) ? 1.f : 0.f;
}
__device__ __forceinline__ double wt_sig(double c, double tau, double taud, int soft)
{
    return wt_sig64(c, tau, soft);
}

// sdev_loc from the two smoothed moments: vari = conv(I^2) - conv(I)^2 ; <= 0 -> 1e-20
// (watroo/wavelets.py:25-28), optional sqrt, then the two factors of wavelets.py:434-436.
// Shared by the variance chain kernel and the bilateral kernel so both give identical bits.
__device__ __forceinline__ float wt_var_point(float p, float m, float f1, float f2, int take_sqrt)
{
#pragma clang fp contract(off)
    float t = p - m * m;
    t = t <= 0.f ? 1e-20f : t;
    if (take_sqrt) t = sqrtf(t);
    return (t * f1) * f2;
}
// (the expression of wt64_var_kernel: identical bits to the two-plane form of the float64 engine)
__device__ __forceinline__ double wt_var_point(double p, double m, double f1, double f2, int take_sqrt)
{
#pragma clang fp contract(off)
    double t = p - m * m;
    t = t <= 0.0 ? 1e-20 : t;
    if (take_sqrt) t = sqrt(t);
    return (t * f1) * f2;
}

// wow per-scale update of one coefficient - watroo/utils.py:193-203:
//   c <- c * significance ; gamma += c ; c <- c * (factor / sqrt(clip(power)))
// Shared by wt_wow_kernel and the fused MODE_WOW chain kernel (identical bits).
__device__ __forceinline__ float wt_wow_point(float c, float power, bool has_power, float nn,
                                              double tau, float tauf, int soft, float factor,
                                              float &gamma_acc)
{
#pragma clang fp contract(off)
    float t = c;
    if (tau > 0.0) t = t * wt_sig(t, tauf * nn, tau * (double)nn, soft);
    gamma_acc = gamma_acc + t;
    float q = factor;
    if (has_power) {
        const float lp = power <= 0.f ? 1e-15f : power;   // utils.py:195
        // factor / sqrt(lp) (utils.py:196,203) as factor * rsq(lp): v_rsq_f32 is 1 ulp, well
        // inside the wow tolerance, and saves the IEEE sqrt + divide sequences (~40 VALU per
        // pixel, which is what made the fused wow kernel 1.5x slower than its filter alone)
        q = factor * __builtin_amdgcn_rsqf(lp);
    }
    return t * q;
}
// float64: the same update; factor / sqrt(lp) as factor * wt_rsq64(lp) (1 ulp; the float64 parity bound
// is 1e-12), the significance by wt_erf64
__device__ __forceinline__ double wt_wow_point(double c, double power, bool has_power, double nn,
                                               double tau, double tauf, int soft, double factor,
                                               double &gamma_acc)
{
#pragma clang fp contract(off)
    double t = c;
    if (tau > 0.0) t = t * wt_sig64(t, tau * nn, soft);
    gamma_acc = gamma_acc + t;
    double q = factor;
    if (has_power) {
        const double lp = power <= 0.0 ? 1e-15 : power;   // utils.py:195
        q = factor * wt_rsq64(lp);
    }
    return t * q;
}

// ---------------------------------------------------------------------------------------------
// K1  generic per-scale separable dilated convolution ("chain march")
//
//   c_{s+1} = h^(s) (*) c_s ,  w_s = c_s - c_{s+1}          watroo/wavelets.py:432,442
//   h^(s) = zero-stuffed outer product of the 1-D taps         watroo/wavelets.py:191-197
//
// A thread owns one group of adjacent columns (4 floats / 2 doubles) and one POLYPHASE ROW CHAIN
// y = q, q+d, q+2d, ...  (d = 2^s): along a chain the dilated vertical filter is an ordinary K-tap
// sliding window that lives in registers, so every input row is fetched once per chain (plus K-1 warm-up
// rows per chunk of S chain steps).  The horizontal taps are K coalesced 16-byte row loads at x + j*d
// (served by L1/L2 after the first touch; for d smaller than the group three aligned loads are recombined
// in registers).  Works for any dilation and any image size (multi-bounce reflection), which is what the
// large scales of wow() (d up to 1024) need; the fused kernels in wt_fused.h take over for the
// small dilations of the headline path.
// ---------------------------------------------------------------------------------------------
// MODE_WOW_PLAIN: the wow update WITHOUT a per-pixel noise map and without the gamma accumulator - the
// common case (scalar noise, h = 0).  A separate instantiation, not a run-time test: conditional loads in
// the row loop make the compiler's vmcnt bookkeeping inexact, every use of a (possibly) loaded value
// then waits for ALL outstanding memory operations - the prefetched rows and the row just stored
// included (13 % of the lattice kernel, 3 % of the row kernel at 8192^2).
// MODE_WOW_GAMMA: scalar noise WITH the gamma accumulator (wow(h > 0)): its row is read with one
// unconditional 16-byte load per lane (clamped column, like the input rows) instead of guarded scalars.
enum { MODE_SMOOTH = 0, MODE_SMOOTH_SQ = 1, MODE_DECOMP = 2, MODE_VAR = 3, MODE_WOW = 4, MODE_WOW_PLAIN = 5, MODE_WOW_GAMMA = 6 };
#define WT_IS_WOW(M) ((M) == MODE_WOW || (M) == MODE_WOW_PLAIN || (M) == MODE_WOW_GAMMA)

// component access of a lane's group
__device__ __forceinline__ void wt_vunpack(float4 v, float (&e)[4]) { e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w; }
__device__ __forceinline__ void wt_vunpack(double2 v, double (&e)[2]) { e[0] = v.x; e[1] = v.y; }
__device__ __forceinline__ float4 wt_vpack(const float (&e)[4]) { return make_float4(e[0], e[1], e[2], e[3]); }
__device__ __forceinline__ double2 wt_vpack(const double (&e)[2]) { return make_double2(e[0], e[1]); }
__device__ __forceinline__ double2 f4_mul(double2 a, double2 b) { return make_double2(a.x * b.x, a.y * b.y); }
__device__ __forceinline__ float wt_fma_s(float a, float b, float c) { return fmaf(a, b, c); }
__device__ __forceinline__ double wt_fma_s(double a, double b, double c) { return fma(a, b, c); }

// Raw operands of the horizontal K-tap filter of one row at the thread's PX pixels:
//   d >= PX: K groups at x + (j-hw) d      d < PX: 3 groups covering x-PX .. x+2PX-1
template <typename T, int K, bool SMALL_D>
__device__ __forceinline__ void wt_hrow_load(const T *row, int x, int d, int W, typename WtVec<T>::V (&raw)[K],
                                             int border = 0)
{
    constexpr int hw = K / 2;
    constexpr int PX = WtVec<T>::PX;
    if constexpr (!SMALL_D) {
#pragma unroll
        for (int j = 0; j < K; ++j) raw[j] = wt_loadv_b(row, x + (j - hw) * d, W, d, border);
    } else {
#pragma unroll
        for (int j = 0; j < 3; ++j) raw[j] = wt_loadv_b(row, x - PX + PX * j, W, d, border);
    }
}

// Horizontal K-tap filter from the raw operands.
//   h   = sum_j k_j v(x + (j-hw) d)                  (v squared first for MODE_SMOOTH_SQ)
//   h2  = sum_j k_j v^2                              (MODE_VAR only)
//   cen = v(x)                                       (centre pixels, for the detail plane)
template <typename T, int K, int MODE, bool SMALL_D>
__device__ __forceinline__ void wt_hrow_filter(const typename WtVec<T>::V *raw, int d, typename WtVec<T>::V &h,
                                               typename WtVec<T>::V &h2, typename WtVec<T>::V &cen)
{
    typedef typename WtVec<T>::V V;
    constexpr int hw = K / 2;
    constexpr int PX = WtVec<T>::PX;
    if constexpr (!SMALL_D) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            V v = raw[j];
            if (j == hw) cen = v;
            V vv = f4_mul(v, v);
            if (MODE == MODE_SMOOTH_SQ || WT_IS_WOW(MODE)) v = vv;
            h = (j == 0) ? f4_scale(wt_tap_s<K, T>(0), v) : f4_fma(wt_tap_s<K, T>(j), v, h);
            if (MODE == MODE_VAR)
                h2 = (j == 0) ? f4_scale(wt_tap_s<K, T>(0), vv) : f4_fma(wt_tap_s<K, T>(j), vv, h2);
        }
    } else {
        // d < PX (float: 1 or 2; double: 1): pixels x-PX .. x+2PX-1 cover every tap (hw*d <= PX)
        const V L = raw[0], C = raw[1], R = raw[2];
        T e[3 * PX], e2[3 * PX];
        {
            T t0[PX], t1[PX], t2[PX];
            wt_vunpack(L, t0);
            wt_vunpack(C, t1);
            wt_vunpack(R, t2);
#pragma unroll
            for (int i = 0; i < PX; ++i) {
                e[i] = t0[i];
                e[PX + i] = t1[i];
                e[2 * PX + i] = t2[i];
            }
        }
#pragma unroll
        for (int i = 0; i < 3 * PX; ++i) {
            e2[i] = e[i] * e[i];
            if (MODE == MODE_SMOOTH_SQ || WT_IS_WOW(MODE)) e[i] = e2[i];
        }
        cen = C;
        T o[PX], o2[PX];
        // d is wave-uniform at run time: one scalar branch, bodies with static indices (selecting per
        // tap with `d == 1 ? e[..] : e[..]` doubled the VALU count of the d < 4 kernels: a v_cndmask
        // per operand)
        auto taps = [&](auto dtag) {
            constexpr int DD = decltype(dtag)::value;
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                T a = wt_tap_s<K, T>(0) * e[PX + k - DD * hw];
                T a2 = wt_tap_s<K, T>(0) * e2[PX + k - DD * hw];
#pragma unroll
                for (int j = 1; j < K; ++j) {
                    a = wt_fma_s(wt_tap_s<K, T>(j), e[PX + k + DD * (j - hw)], a);
                    a2 = wt_fma_s(wt_tap_s<K, T>(j), e2[PX + k + DD * (j - hw)], a2);
                }
                o[k] = a;
                o2[k] = a2;
            }
        };
        if constexpr (PX == 4) {
            if (d == 1) taps(std::integral_constant<int, 1>{});
            else taps(std::integral_constant<int, 2>{});
        } else {
            taps(std::integral_constant<int, 1>{});
        }
        h = wt_vpack(o);
        if (MODE == MODE_VAR) h2 = wt_vpack(o2);
    }
}

// XCD-aware block remap for the chain-march kernels.  Hardware deals consecutive workgroup ids
// round-robin over the 8 XCDs (each with a private 4 MiB L2).  The horizontal taps of a
// dilated filter re-read the SAME image rows at x +- d, x +- 2d, i.e. from the x-blocks next
// to this one; with the default order those neighbours sit on other XCDs and every XCD
// fetches the row segment again through the fabric.  Remapped, all x-blocks of one group of
// rows run on one XCD and the re-reads are L2 hits.  Pure speed: any placement is correct.
// Requires gridDim.y % 8 == 0 (the host rounds up; surplus blocks find no work and exit).
__device__ __forceinline__ void wt_xcd_remap(int &bx, int &by)
{
    const int gx = gridDim.x;
    const int b = blockIdx.y * gx + blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    bx = j % gx;
    by = (j / gx) * 8 + xcd;
}

template <typename T>
struct ChainArgsT {
    const T *in;  // local row 0 of the input plane
    T *out_c;     // smooth / variance output (local row 0)
    T *out_w;     // detail output or nullptr
    const T *aux; // bilateral: per-pixel variance plane
    Geo g;        // (P = row pitch in elements of T)
    int d;        // dilation 2^s
    int S;        // chain steps per thread
    int chunks;   // chunks per chain
    T f1, f2;     // MODE_VAR: factors applied to the clipped variance (wavelets.py:434-436)
    int take_sqrt;
    // MODE_WOW (fused wow per-scale update) / bilateral with in-kernel variance
    const T *noise;      // per-pixel noise map or nullptr
    T *gamma;            // gamma accumulator plane or nullptr
    double tau;          // significance threshold (<= 0: none)
    T factor;            // w * power_norm
    int soft, whiten, inline_var;
    int nt;              // streaming stores for the outputs (planes >> cache)
};
typedef ChainArgsT<float> ChainArgs;

// Vertical half shared by the chain-march kernel (taps fetched from global memory) and the
// row kernel (taps fetched from an LDS copy of the row): horizontal filter of each incoming row,
// K-row sliding window of the filtered rows, vertical filter, mode epilogue.  One code path so
// that both kernels produce identical bits.
template <typename T, int K, int MODE, bool SMALL_D>
struct WtVert {
    typedef typename WtVec<T>::V V;
    static constexpr int PX = WtVec<T>::PX;
    static constexpr int hw = K / 2;
    V hwin[K], h2win[K], cen[hw + 1];

    // warm-up rows r0-hw .. r0+hw-1 (j = 0 .. K-2)
    __device__ __forceinline__ void prime(int j, const V *raw, int d)
    {
        V ct;
        hwin[j] = wt_vzero<V>();
        h2win[j] = wt_vzero<V>();
        wt_hrow_filter<T, K, MODE, SMALL_D>(raw, d, hwin[j], h2win[j], ct);
        if (j >= hw) cen[j - hw] = ct;
    }

    // row r+hw enters; emits row r of the outputs at element offset `off` (row start), pixel x
    __device__ __forceinline__ void emit(const V *raw, const ChainArgsT<T> &a, int64_t off, int x,
                                         bool lane_ok)
    {
        const Geo &g = a.g;
        hwin[K - 1] = wt_vzero<V>();
        h2win[K - 1] = wt_vzero<V>();
        wt_hrow_filter<T, K, MODE, SMALL_D>(raw, a.d, hwin[K - 1], h2win[K - 1], cen[hw]);
        V o = f4_scale(wt_tap_s<K, T>(0), hwin[0]);
#pragma unroll
        for (int j = 1; j < K; ++j) o = f4_fma(wt_tap_s<K, T>(j), hwin[j], o);
        if (MODE == MODE_VAR) {
            V p = f4_scale(wt_tap_s<K, T>(0), h2win[0]);
#pragma unroll
            for (int j = 1; j < K; ++j) p = f4_fma(wt_tap_s<K, T>(j), h2win[j], p);
            T pp[PX], mm[PX], v[PX];
            wt_vunpack(p, pp);
            wt_vunpack(o, mm);
#pragma unroll
            for (int k = 0; k < PX; ++k) v[k] = wt_var_point(pp[k], mm[k], a.f1, a.f2, a.take_sqrt);
            wt_storev(a.out_c + off, x, g.P, wt_vpack(v), a.nt, lane_ok);
        } else if (WT_IS_WOW(MODE)) {
            // fused wow update: o = conv_s(c^2) (local power), cen[0] = c at this row; result
            // goes to a different plane (the host swaps plane pointers afterwards)
            T cc[PX], pw[PX], nn[PX], gg[PX], r4[PX];
            wt_vunpack(cen[0], cc);
            wt_vunpack(o, pw);
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                nn[k] = (T)1;
                gg[k] = (T)0;
            }
            if constexpr (MODE == MODE_WOW_GAMMA) {
                // (columns clamped into the row's pitch: lanes that store nothing read something harmless)
                const V g4 = *reinterpret_cast<const V *>(a.gamma + off + min(max(x, 0), g.P - PX));
                wt_vunpack(g4, gg);
            }
            if constexpr (MODE == MODE_WOW) {
                const bool full = x + PX - 1 < g.W;
                if (a.noise && lane_ok) {
#pragma unroll
                    for (int k = 0; k < PX; ++k) if (full || x + k < g.W) nn[k] = a.noise[off + x + k];
                }
                if (a.gamma && lane_ok) {
#pragma unroll
                    for (int k = 0; k < PX; ++k) if (full || x + k < g.W) gg[k] = a.gamma[off + x + k];
                }
            }
            const T tauf = (T)a.tau;
#pragma unroll
            for (int k = 0; k < PX; ++k)
                r4[k] = wt_wow_point(cc[k], pw[k], a.whiten != 0, nn[k], a.tau, tauf, a.soft, a.factor, gg[k]);
            wt_storev(a.out_c + off, x, g.P, wt_vpack(r4), a.nt, lane_ok);
            if constexpr (MODE == MODE_WOW) {
                if (a.gamma) wt_storev(a.gamma + off, x, g.P, wt_vpack(gg), a.nt, lane_ok);
            }
            if constexpr (MODE == MODE_WOW_GAMMA) wt_storev(a.gamma + off, x, g.P, wt_vpack(gg), a.nt, lane_ok);
        } else {
            wt_storev(a.out_c + off, x, g.P, o, a.nt, lane_ok);
            if (MODE == MODE_DECOMP && a.out_w)
                wt_storev(a.out_w + off, x, g.P, f4_sub(cen[0], o), a.nt, lane_ok);
        }
#pragma unroll
        for (int j = 0; j < K - 1; ++j) {
            hwin[j] = hwin[j + 1];
            h2win[j] = h2win[j + 1];
        }
#pragma unroll
        for (int j = 0; j < hw; ++j) cen[j] = cen[j + 1];
    }
};

template <typename T, int K, int MODE, bool SMALL_D>
__global__ __launch_bounds__(256) void wt_chain_kernel(ChainArgsT<T> a)
{
    typedef typename WtVec<T>::V V;
    constexpr int PX = WtVec<T>::PX;
    constexpr int hw = K / 2;
    const Geo g = a.g;
    int bx, by;
    wt_xcd_remap(bx, by);
    const int x = (bx * 64 + threadIdx.x) * PX;
    if (x >= g.W) return;
    // one wave = one threadIdx.y: make the item (and with it the chain phase, the chunk, the row
    // pointers and the loop counters) scalar - the compiler cannot prove threadIdx.y wave-uniform
    const int item = __builtin_amdgcn_readfirstlane(by * blockDim.y + threadIdx.y);
    const int d = a.d;
    const int q = item % d;   // chain phase (local row offset)
    const int c = item / d;   // chunk along the chain
    if (c >= a.chunks || q >= g.nrows) return;
    const int n_q = (g.nrows - q + d - 1) / d;  // chain length
    const int r0 = c * a.S;
    const int r1 = min(r0 + a.S, n_q);
    if (r0 >= r1) return;

    WtVert<T, K, MODE, SMALL_D> vert;
    const int gy0 = g.row0 + q;  // global row of chain element 0
    V raw[K], nxt[K];
#pragma unroll
    for (int j = 0; j < K - 1; ++j) {
        wt_hrow_load<T, K, SMALL_D>(wt_row_b(a.in, g, gy0 + d * (r0 - hw + j), d), x, d, g.W, raw, g.border);
        vert.prime(j, raw, d);
    }
    // software prefetch: the operands of the NEXT chain row are in flight while this row is
    // filtered (the kernel is latency-bound at 3-4 waves/SIMD otherwise)
    wt_hrow_load<T, K, SMALL_D>(wt_row_b(a.in, g, gy0 + d * (r0 + hw), d), x, d, g.W, nxt, g.border);
    for (int r = r0; r < r1; ++r) {
#pragma unroll
        for (int j = 0; j < K; ++j) raw[j] = nxt[j];
        wt_hrow_load<T, K, SMALL_D>(wt_row_b(a.in, g, gy0 + d * min(r + 1, r1 - 1) + d * hw, d), x, d, g.W, nxt, g.border);
        vert.emit(raw, a, (int64_t)(q + d * r) * g.P, x, true);
    }
}

// ---------------------------------------------------------------------------------------------
// K1c  "lattice" kernel: the chain march for LARGE dilations (d >= 256, wow() scales 8-10, where
// the x halo no longer fits a workgroup and the chain kernel pays K tap loads per row).  A thread
// owns C columns of the POLYPHASE LATTICE in x as well - pixel groups x0, x0+d, ..., x0+(C-1)d of its
// chain - so neighbouring lattice columns share taps in registers: C+K-1 row loads feed C
// horizontal filters (2 loads per output for C = 4 instead of 5).  Lanes run over the phase
// (consecutive pixels), so every load is still a coalesced 16 B per lane; reflection is per tap
// address as in the chain kernel, so any width / border mode works.  Arithmetic is WtVert per
// lattice column: bit-identical to the chain and row kernels.
// ---------------------------------------------------------------------------------------------
template <typename T, int K, int MODE, int C>
__global__ __launch_bounds__(256, 2) void wt_lattice_kernel(ChainArgsT<T> a)
{
    typedef typename WtVec<T>::V V;
    constexpr int PX = WtVec<T>::PX;
    constexpr int hw = K / 2;
    constexpr int NR = C + K - 1;                // row operands per step
    const Geo g = a.g;
    int bx, by;
    wt_xcd_remap(bx, by);
    const int d = a.d;
    const int p4 = d >> (PX == 4 ? 2 : 1);       // group phases per lattice column (d % PX == 0)
    const int t = bx * 64 + threadIdx.x;
    const int gi = t / p4, ph = t - gi * p4;
    const int x0 = PX * ph + d * C * gi;         // first lattice column of this thread
    if (x0 >= g.W) return;
    // one wave = one threadIdx.y: make the item (and with it the chain phase, the chunk, the row
    // pointers and the loop counters) scalar - the compiler cannot prove threadIdx.y wave-uniform
    const int item = __builtin_amdgcn_readfirstlane(by * blockDim.y + threadIdx.y);
    const int q = item % d;
    const int c = item / d;
    if (c >= a.chunks || q >= g.nrows) return;
    const int n_q = (g.nrows - q + d - 1) / d;
    const int r0 = c * a.S;
    const int r1 = min(r0 + a.S, n_q);
    if (r0 >= r1) return;

    // The operand columns do not depend on the row: with W % PX == 0 (host-checked) an aligned
    // group of PX pixels is either inside the image or entirely outside, and the symmetric
    // reflection of an outside group is an aligned group read backwards (even number of
    // bounces: forwards).  One offset and one flag per operand, no branches in the row loop.
    int off[NR];
    unsigned rev = 0;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int xo = x0 + (j - hw) * d;
        const int a0 = wt_refl(xo, g.W), a3 = wt_refl(xo + PX - 1, g.W);
        off[j] = min(a0, a3);
        if (a3 < a0) rev |= 1u << j;
    }
    WtVert<T, K, MODE, false> vert[C];
    const int gy0 = g.row0 + q;
    V lat[NR], nxt[NR];
    auto load_lat = [&](int r, V (&dst)[NR]) {
        const T *row = wt_row(a.in, g, gy0 + d * r);
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const V v = *reinterpret_cast<const V *>(row + off[j]);
            dst[j] = (rev >> j) & 1u ? wt_vrev(v) : v;
        }
    };
#pragma unroll
    for (int j = 0; j < K - 1; ++j) {
        load_lat(r0 - hw + j, lat);
#pragma unroll
        for (int cc = 0; cc < C; ++cc) vert[cc].prime(j, lat + cc, d);
    }
    load_lat(r0 + hw, nxt);
    for (int r = r0; r < r1; ++r) {
#pragma unroll
        for (int j = 0; j < NR; ++j) lat[j] = nxt[j];
        load_lat(min(r + 1, r1 - 1) + hw, nxt);
        const int64_t o = (int64_t)(q + d * r) * g.P;
#pragma unroll
        for (int cc = 0; cc < C; ++cc) vert[cc].emit(lat + cc, a, o, x0 + cc * d, x0 + cc * d < g.W);
    }
}

// ---------------------------------------------------------------------------------------------
// K1b  "row" kernel: the same single-scale operators for the dilations whose horizontal halo
// fits a workgroup (hw*d <= 1/8 of its width).  A workgroup of NW waves marches down one chunk
// of one polyphase chain like the fused pass: ONE coalesced 16-byte load per lane per row, the
// row is shared through LDS (double-buffered, one barrier per PAIR of rows) and the K dilated taps are
// LDS reads at lane offsets +-d/PX, +-2d/PX (or the two adjacent lanes for d < PX) - instead of K
// global loads per row.  Arithmetic is WtVert, i.e. bit-identical to the chain kernel.
// ---------------------------------------------------------------------------------------------
template <typename T>
struct RowArgsT {
    ChainArgsT<T> c;
    int Vx;   // valid (stored) pixels per x-strip, multiple of 32
    int HX;   // x halo in pixels, multiple of 32
};
typedef RowArgsT<float> RowArgs;

template <typename T, int K, int MODE, bool SMALL_D, int NW>
__global__ __launch_bounds__(NW * 64) void wt_row_kernel(RowArgsT<T> ra)
{
    typedef typename WtVec<T>::V V;
    constexpr int PX = WtVec<T>::PX;
    constexpr int hw = K / 2;
    constexpr int NL = NW * 64;
    __shared__ V rowbuf[2][2][NL];                       // [pair parity][row of the pair][lane]
    const ChainArgsT<T> &a = ra.c;
    const Geo g = a.g;
    const int d = a.d;
    const int gl = threadIdx.x;
    const int X0 = blockIdx.x * ra.Vx;
    const int x = X0 - ra.HX + PX * gl;
    const int item = blockIdx.y;
    const int q = item % d;
    const int c = item / d;
    if (c >= a.chunks || q >= g.nrows) return;           // whole workgroup exits together
    const int n_q = (g.nrows - q + d - 1) / d;
    const int r0 = c * a.S;
    const int r1 = min(r0 + a.S, n_q);
    if (r0 >= r1) return;

    const bool lane_ok = (x >= X0) && (x < X0 + ra.Vx) && (x < g.W);
    const bool lane_interior = (x >= 0) && (x + PX - 1 < g.W);
    const bool wave_has_edge = !__all(lane_interior);
    const int xc = min(max(x, 0), g.P - PX);
    const int xi0 = wt_refl_b(x, g.W, d, g.border), xi1 = wt_refl_b(x + 1, g.W, d, g.border);
    int xi2 = 0, xi3 = 0;
    if constexpr (PX == 4) {
        xi2 = wt_refl_b(x + 2, g.W, d, g.border);
        xi3 = wt_refl_b(x + 3, g.W, d, g.border);
    }
    const int gy0 = g.row0 + q;
    const int t_last = r1 - 1 + hw;
    auto load_row = [&](int t) -> V {
        const T *row = wt_row_b(a.in, g, gy0 + d * min(t, t_last), d);
        V v = *reinterpret_cast<const V *>(row + xc);
        if (wave_has_edge) {
            if (!lane_interior) {
                if constexpr (PX == 4) v = make_float4(row[xi0], row[xi1], row[xi2], row[xi3]);
                else v = make_double2(row[xi0], row[xi1]);
            }
        }
        return v;
    };
    // taps of this lane out of the shared row (out-of-range lanes clamp: halo lanes only)
    const int lo = d / PX;                               // lane offset of one dilation step
    auto gather = [&](const V *rowv, V own, V (&raw)[K]) {
        if constexpr (SMALL_D) {
            raw[0] = rowv[max(gl - 1, 0)];
            raw[1] = own;
            raw[2] = rowv[min(gl + 1, NL - 1)];
        } else {
#pragma unroll
            for (int j = 0; j < K; ++j)
                raw[j] = (j == hw) ? own : rowv[min(max(gl + (j - hw) * lo, 0), NL - 1)];
        }
    };

    WtVert<T, K, MODE, SMALL_D> vert;
    V raw[K];
    // Four rows in flight, in NAMED registers used in turn (the loop is unrolled by that many): a
    // rotating array (pf0 = pf1; pf1 = load) makes the compiler copy the load it has just issued at
    // the end of every iteration, i.e. wait for it at once - no prefetch left.
    // TWO ROWS PER BARRIER (round 3): the kernel sat at s_waitcnt / s_barrier for 72 % of its wave
    // cycles with one barrier per row (SQ_WAIT_ANY, profiles/r02_e) - four waves re-synchronising
    // every ~130 VALU instructions.  A step now shares a PAIR of rows through LDS (two row buffers
    // per parity) behind one barrier and filters both; same arithmetic per row, identical bits.
    V pfa = load_row(r0 - hw), pfb = load_row(r0 - hw + 1);
    V pfc = load_row(r0 - hw + 2), pfd = load_row(r0 - hw + 3);
    // steps t = r0-hw .. r1-1+hw ; the pair index selects the LDS buffers
    const int nsteps = (r1 - r0) + 2 * hw;
    auto share = [&](const int k, const V cur0, const V cur1, const V *&rv0, const V *&rv1) {
        V *w0 = rowbuf[(k >> 1) & 1][0], *w1 = rowbuf[(k >> 1) & 1][1];
        w0[gl] = cur0;
        w1[gl] = cur1;
        __syncthreads();
        rv0 = w0;
        rv1 = w1;
    };
    auto emit_row = [&](const int k, const V *rowv, const V cur) {
        gather(rowv, cur, raw);
        vert.emit(raw, a, (int64_t)(q + d * (r0 - 2 * hw + k)) * g.P, x, lane_ok);   // row t - hw, t = r0 - hw + k
    };
    auto pair_emit = [&](const int k, const V cur0, const V cur1) {
        const V *rv0, *rv1;
        share(k, cur0, cur1, rv0, rv1);
        emit_row(k, rv0, cur0);
        if (k + 1 < nsteps) emit_row(k + 1, rv1, cur1);      // workgroup-uniform
    };
    // warm-up rows k = 0 .. K-2 fill the window: (K-1)/2 pairs with STATIC window indices (a switch on
    // the run-time step number made the compiler index the window dynamically: scratch memory)
    {
        const V c0 = pfa, c1 = pfb;
        pfa = load_row(r0 - hw + 4);
        pfb = load_row(r0 - hw + 5);
        const V *rv0, *rv1;
        share(0, c0, c1, rv0, rv1);
        gather(rv0, c0, raw);
        vert.prime(0, raw, d);
        gather(rv1, c1, raw);
        vert.prime(1, raw, d);
    }
    {
        const V c0 = pfc, c1 = pfd;
        pfc = load_row(r0 - hw + 6);
        pfd = load_row(r0 - hw + 7);
        if constexpr (K > 3) {
            const V *rv0, *rv1;
            share(2, c0, c1, rv0, rv1);
            gather(rv0, c0, raw);
            vert.prime(2, raw, d);
            gather(rv1, c1, raw);
            vert.prime(3, raw, d);
        } else {
            pair_emit(2, c0, c1);                            // (nsteps >= 3: row 2 exists)
        }
    }
    for (int k = 4; k < nsteps; k += 4) {
        {
            const V c0 = pfa, c1 = pfb;
            pfa = load_row(r0 - hw + k + 4);
            pfb = load_row(r0 - hw + k + 5);
            pair_emit(k, c0, c1);
        }
        if (k + 2 < nsteps) {                                // workgroup-uniform
            const V c0 = pfc, c1 = pfd;
            pfc = load_row(r0 - hw + k + 6);
            pfd = load_row(r0 - hw + k + 7);
            pair_emit(k + 2, c0, c1);
        }
    }
}
