// Device helpers shared by every translation unit of libwatroo_hip.so (the per-scale kernels of
// wt_kernels_*.h and the fused passes of wt_fused.h, which are compiled one instantiation group per
// translation unit - wt_fused_tu.hip): border reflection, the scaling functions' taps, float4
// arithmetic.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include "wt_internal.h"

#define WT_HIST_BINS 2048      // bins of one radix level of the exact-median select (wt_kernels_apps.h)

// ---------------------------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------------------------
// Symmetric reflection with edge duplication and any number of bounces:
// cv2.BORDER_REFLECT (watroo/wavelets.py:45) == np.pad 'symmetric' (watroo/wavelets.py:77).
__device__ __forceinline__ int wt_refl(int i, int n)
{
    if ((unsigned)i < (unsigned)n) return i;
    const int p = 2 * n;
    int m = i % p;
    if (m < 0) m += p;
    return m < n ? m : p - 1 - m;
}

// Border rule of the reference's RECURSIVE algorithm (watroo/wavelets.py:354-390): each
// polyphase sub-array (offset o, stride d) is filtered on its own with BORDER_REFLECT, i.e. an
// out-of-range index reflects inside its own residue class:  i = o + d j  ->  o + d refl(j, n_o).
__device__ __forceinline__ int wt_refl_b(int i, int n, int d, int border)
{
    if ((unsigned)i < (unsigned)n) return i;
    if (border == 0) return wt_refl(i, n);
    if (border == 2) {
        // scipy.ndimage 'mirror' (1-D branch of convolution(), watroo/wavelets.py:66-69):
        // reflection about the centre of the edge sample, d c b | a b c d | c b a
        if (n == 1) return 0;
        const int p = 2 * n - 2;
        int m = i % p;
        if (m < 0) m += p;
        return m < n ? m : p - m;
    }
    int o = i % d;
    if (o < 0) o += d;
    const int j = (i - o) / d;
    const int n_o = (n - o + d - 1) / d;
    if (border == 3) {
        // 'mirror' inside the residue class: the 1-D branch of convolution() applied to the
        // sub-arrays of atrous_recursive (watroo/wavelets.py:66-69 under :354-390)
        if (n_o == 1) return o;
        const int p = 2 * n_o - 2;
        int m = j % p;
        if (m < 0) m += p;
        return o + d * (m < n_o ? m : p - m);
    }
    return o + d * wt_refl(j, n_o);
}

template <int K>
__device__ __forceinline__ constexpr float wt_tap(int i)
{
    // Triangle (watroo/wavelets.py:239) and B3spline (watroo/wavelets.py:268) 1-D taps:
    // dyadic rationals, exact in fp32.
    if (K == 3) return i == 1 ? 0.5f : 0.25f;
    return (i == 2) ? 0.375f : ((i == 1 || i == 3) ? 0.25f : 0.0625f);
}

template <int K>
__device__ __forceinline__ constexpr float wt_tap_log2(int i)
{
    // log2 of the taps: 1/4, 1/2 (Triangle); 1/16, 1/4, 3/8 (B3spline)
    if (K == 3) return i == 1 ? -1.f : -2.f;
    return (i == 2) ? -1.4150374992788438f : ((i == 1 || i == 3) ? -2.f : -4.f);
}

__device__ __forceinline__ float4 f4_mul(float4 a, float4 b)
{
    return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}
__device__ __forceinline__ float4 f4_scale(float k, float4 a)
{
    return make_float4(k * a.x, k * a.y, k * a.z, k * a.w);
}
__device__ __forceinline__ float4 f4_fma(float k, float4 a, float4 c)
{
    return make_float4(fmaf(k, a.x, c.x), fmaf(k, a.y, c.y), fmaf(k, a.z, c.z), fmaf(k, a.w, c.w));
}
__device__ __forceinline__ float4 f4_add(float4 a, float4 b)
{
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b)
{
    return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
}

// a / b for the two divisions of the bilateral kernels (b = the weight sum in [k_c, 1], or a
// variance >= 1e-20): v_rcp_f32, one Newton step on the reciprocal and one residual correction of
// the quotient - 6 instructions instead of the ~10 of the IEEE sequence (v_div_scale x2, v_rcp,
// 4 fma, v_div_fmas, v_div_fixup), the same result except for rare 1-ulp cases (no scaling is
// needed: neither operand is near the ends of the exponent range).  Bilateral outputs are a
// stated-tolerance path (2e-5 * max|input|), not a bit-exact one.
__device__ __forceinline__ float wt_div_nr(float a, float b)
{
    float r = __builtin_amdgcn_rcpf(b);
    r = fmaf(fmaf(-b, r, 1.0f), r, r);
    const float q = a * r;
    return fmaf(fmaf(-b, q, a), r, q);
}

// Element type of a pass (round 3): float - a lane owns 4 adjacent pixels - or double - 2 pixels.
// Either way a lane moves 16 bytes per access and the register window costs the same VGPRs, so the
// float64 passes are the float32 design at twice the bytes per pixel (the reference computes
// float64 / integer inputs in float64, watroo/wavelets.py:297,319-320).
template <typename T> struct WtVec;
template <> struct WtVec<float> {
    typedef float4 V;
    static constexpr int PX = 4;
};
template <> struct WtVec<double> {
    typedef double2 V;
    static constexpr int PX = 2;
};
__device__ __forceinline__ double2 f4_scale(double k, double2 a) { return make_double2(k * a.x, k * a.y); }
__device__ __forceinline__ double2 f4_fma(double k, double2 a, double2 c) { return make_double2(fma(k, a.x, c.x), fma(k, a.y, c.y)); }
__device__ __forceinline__ double2 f4_add(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 f4_sub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
// the taps are dyadic rationals: exact in either precision
template <int K, typename T>
__device__ __forceinline__ constexpr T wt_tap_s(int i) { return (T)wt_tap<K>(i); }
template <typename V> __device__ __forceinline__ V wt_vzero();
template <> __device__ __forceinline__ float4 wt_vzero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ __forceinline__ double2 wt_vzero<double2>() { return make_double2(0.0, 0.0); }
// the group read backwards (reflection of an aligned group that lies outside the image)
__device__ __forceinline__ float4 wt_vrev(float4 v) { return make_float4(v.w, v.z, v.y, v.x); }
__device__ __forceinline__ double2 wt_vrev(double2 v) { return make_double2(v.y, v.x); }
// the group that straddles the right image border (W % PX = r != 0): symmetric reflection of its own pixels; for
// r == 1 the loaded group is the four pixels that END at the border
__device__ __forceinline__ float4 wt_vstraddle(float4 v, int r)
{
    return r == 2 ? make_float4(v.x, v.y, v.y, v.x) : (r == 3 ? make_float4(v.x, v.y, v.z, v.z) : make_float4(v.w, v.w, v.z, v.y));
}
__device__ __forceinline__ double2 wt_vstraddle(double2 v, int) { return make_double2(v.x, v.x); }
// scheduling fence on the components of a row (see the FAST path of the kernel)
__device__ __forceinline__ void wt_vfence(float4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void wt_vfence(double2 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
