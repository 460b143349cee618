// Double-precision building blocks of the float64 kernels (gfx950 has no double-precision erf / exp /
// fast rsqrt): wt_erf64, wt_exp2_64, wt_rsq64, wt_div64 and the significance forms built on them.
//
// The polynomial tables live in __constant__ memory WITH EXTERNAL LINKAGE on purpose: a 64-bit literal
// cannot be an operand of v_fma_f64, and for a table the compiler can see through (constexpr, or a static
// __constant__ nobody writes) it folds the coefficients back into the instruction stream - a v_mov_b64 +
// v_fmac_f64 pair per Horner step; from an external table they arrive by scalar loads in SGPR pairs,
// which v_fma_f64 reads directly.  Every translation unit that includes this header defines its own
// tables (no relocatable device code: a code object per unit), so the host-side names must differ per
// unit: the build passes -DWT_TU_NAME=<unit>.
#pragma once
#include <hip/hip_runtime.h>

#ifndef WT_TU_NAME
#define WT_TU_NAME api
#endif
#define WT_PASTE2(a, b) a##_##b
#define WT_PASTE(a, b) WT_PASTE2(a, b)
#define WT_ERF64_Q WT_PASTE(WT_ERF64_Q, WT_TU_NAME)
#define WT_ERF64_F WT_PASTE(WT_ERF64_F, WT_TU_NAME)
#define WT_EXP2_64_C WT_PASTE(WT_EXP2_64_C, WT_TU_NAME)

// erf(y) for y >= 0 in double precision, branch-free (round 4).  The library erf evaluates one of
// several ranges per lane - a wave pays for all of them, ~200 double-precision operations per sample,
// which bounded the float64 threshold kernels at 0.40 of the HBM rate.  Here
//     erf(y) = -expm1(a),   a = -y * (r(t) + y) = log(erfc(y)),   t = y / 3 - 1,   y clamped to 6
// (erfc(6) = 2e-17), r ONE degree-26 polynomial (tools/make_erf64.py: weighted Chebyshev fit against
// 50-digit mpmath values), and expm1 written out: a = n ln2 + x, |x| <= ln2 / 2,
// expm1(a) = 2^n (1 + x q(x)) - 1 with q of degree 12, so that -expm1(a) = (1 - 2^n) - 2^n x q(x) in one
// FMA - exact for n = 0, i.e. small arguments keep their relative accuracy.  ~50 FMAs per sample, no
// division, no branch.  Measured against mpmath on [1e-300, 6.5]: absolute error <= 2.3e-16, relative
// <= 2.7e-14 (the float64 parity bound of the tests is 1e-12).  erf(0) = 0 exactly; NaN stays NaN.
// The coefficients live in constant memory, NOT in the instruction stream: a 64-bit literal cannot be
// an operand of v_fma_f64, so with constexpr tables every Horner step was a v_mov_b64 + v_fmac_f64
// pair (150 moves per two samples); scalar loads put them in SGPR pairs, which v_fma_f64 reads directly.
__constant__ double WT_ERF64_Q[27] = {0x1.259bcee7098c9p-1, -0x1.142c68ccd863dp-2, 0x1.2293b824c872dp-3, -0x1.33ec0d4b613eap-4,
                                      0x1.3c837774a376ap-5, -0x1.32f7d405cccfap-6, 0x1.0ee44d87ebd23p-7, -0x1.93aca6b315f22p-9,
                                      0x1.84922d81926a0p-11, 0x1.3fd374d9a6dcdp-13, -0x1.8d18e3e4439d9p-12, 0x1.63d73d5b4b8bap-12,
                                      -0x1.e1bf08c08a3fap-13, 0x1.22e8a9c630114p-13, -0x1.19446361f8e4fp-14, -0x1.49050d2260ba0p-22,
                                      0x1.86930026aa01cp-20, 0x1.287c357a01b50p-15, 0x1.14679d6bb6ed4p-16, -0x1.d2963129dd6bep-15,
                                      -0x1.4bd1e396300c0p-16, 0x1.2e935cbc2a44fp-15, 0x1.8e87fa9480c7fp-16, -0x1.122db495799f0p-16,
                                      -0x1.813c9c90f35cap-17, 0x1.c52e5511ca2c2p-19, 0x1.147aa0cdd9e8cp-19};
// 1 / k!, k = 1 .. 13
__constant__ double WT_ERF64_F[13] = {1.0, 1.0 / 2, 1.0 / 6, 1.0 / 24, 1.0 / 120, 1.0 / 720, 1.0 / 5040, 1.0 / 40320, 1.0 / 362880,
                                      1.0 / 3628800, 1.0 / 39916800, 1.0 / 479001600, 1.0 / 6227020800.0};
__device__ __forceinline__ double wt_erf64(double y)
{
    const double *Q = WT_ERF64_Q, *F = WT_ERF64_F;
    y = y > 6.0 ? 6.0 : y;
    const double t = fma(y, 1.0 / 3.0, -1.0);
    double r = Q[26];
#pragma unroll
    for (int k = 25; k >= 0; --k) r = fma(r, t, Q[k]);
    const double a = -(y * (r + y));                         // log(erfc(y)), in [-38.5, 0]
    const double n = __builtin_rint(a * 0x1.71547652b82fep+0);
    double x = fma(n, -0x1.62e42fefa39efp-1, a);
    x = fma(n, -0x1.abc9e3b39803fp-56, x);
    double q = F[12];
#pragma unroll
    for (int k = 11; k >= 0; --k) q = fma(q, x, F[k]);
    const double s = ldexp(1.0, (int)n);
    return fma(-s, x * q, 1.0 - s);
}

// significance of one sample (watroo/wavelets.py:137-141): erf(|v / tt|) or |v| > tt
__device__ __forceinline__ double wt_sig64(double v, double tt, int soft)
{
    return soft ? wt_erf64(fabs(v / tt)) : (fabs(v) > tt ? 1.0 : 0.0);
}
// the same with the reciprocal of a scalar threshold (the kernels form 1 / tau once per thread: the
// argument of erf then differs from the quotient by at most one rounding, 1e-16 relative)
__device__ __forceinline__ double wt_sig64_inv(double v, double tt, double inv_tt, int soft)
{
    return soft ? wt_erf64(fabs(v) * inv_tt) : (fabs(v) > tt ? 1.0 : 0.0);
}

// ---------------------------------------------------------------------------------------------
// double-precision building blocks of the float64 per-scale kernels (wt_stencil.h, round 5)
// ---------------------------------------------------------------------------------------------
// 1 / sqrt(x), x > 0 and far from the ends of the exponent range (the clipped local power of wow,
// watroo/utils.py:195-196): v_rsq_f64 (about 2^-26) and two Newton steps, ~1 ulp - instead of the IEEE
// sqrt and division sequences (~60 double-precision instructions per pixel).
__device__ __forceinline__ double wt_rsq64(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = fma(y, fma(-hx * y, y, 0.5), y);
    y = fma(y, fma(-hx * y, y, 0.5), y);
    return y;
}

// a / b for the divisions of the float64 bilateral kernel (b = the weight sum in [k_c, 1], or a variance
// >= 1e-20): v_rcp_f64, two Newton steps on the reciprocal, one residual correction of the quotient.
__device__ __forceinline__ double wt_div64(double a, double b)
{
    double r = __builtin_amdgcn_rcp(b);
    r = fma(fma(-b, r, 1.0), r, r);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}

// 2^t for t <= 0 (the range weights of the bilateral filter, watroo/wavelets.py:97, in base 2): there is
// no double-precision exponential instruction.  t = n + f, n = round(t) taken from the low mantissa bits
// of t + 1.5 * 2^52, |f| <= 1/2, 2^f by a degree-10 polynomial (tools/make_exp2_64.py: Chebyshev
// interpolant of 2^f at 50 digits, 4.4e-16 relative in double Horner form), 2^n added into the exponent
// field.  t is clamped at -1020 (2^-1020 = 9e-308: nothing in a weight sum of at least k_c sees it), so
// the result never leaves the normal range; NaN arguments give 2^-1020.  15 double-precision operations
// and one integer add.  The coefficients live in constant memory for the reason given at wt_erf64.
__constant__ double WT_EXP2_64_C[11] = {0x1.0000000000000p+0, 0x1.62e42fefa3a19p-1, 0x1.ebfbdff82c598p-3, 0x1.c6b08d703ce49p-5,
                                               0x1.3b2ab6fba1ddap-7, 0x1.5d87fe9d7a584p-10, 0x1.430913096fd9fp-13, 0x1.ffcb54062e698p-17,
                                               0x1.62bfd47773353p-20, 0x1.b675bca4eeebbp-24, 0x1.e6063f7217bc6p-28};
// The form of the bilateral kernel (wt_bilateral64.h), whose exponent is an FMA: with
//   u = fma(delta^2, s / 64, 1 + log2(k) / 64)   clamped to [0, 1] by the FMA's own output modifier
// (t = 64 (u - 1) floored at -64: weights below 2^-64 = 5e-20 of the centre tap count as 2^-64; no v_max_f64),
// m = u + 1.5 * 2^46 carries round(64 u) = round(t) + 64 in its low mantissa bits, g = u - (m - 1.5 * 2^46) =
// (t - round(t)) / 64, and 2^(64 g - 64) is the SAME polynomial with its coefficients scaled by 2^(6 k - 64) -
// powers of two: every Horner step is the unscaled step times a power of two, bit for bit.  The exponent
// field then takes round(t) + 64 as it stands.  wt_exp2u_split: the range reduction; wt_exp2u_join: the
// exponent.  16 double-precision operations and one integer add per weight, the exponent FMA included; t is
// quantised to 64 ulp(1) = 7e-15 (relative error of the result 5e-15).
#define WT_EXP2U_C WT_PASTE(WT_EXP2U_C, WT_TU_NAME)
__constant__ double WT_EXP2U_C[11] = {0x1.0000000000000p-64, 0x1.62e42fefa3a19p-59, 0x1.ebfbdff82c598p-55, 0x1.c6b08d703ce49p-51,
                                      0x1.3b2ab6fba1ddap-47, 0x1.5d87fe9d7a584p-44, 0x1.430913096fd9fp-41, 0x1.ffcb54062e698p-39,
                                      0x1.62bfd47773353p-36, 0x1.b675bca4eeebbp-34, 0x1.e6063f7217bc6p-32};
__device__ __forceinline__ void wt_exp2u_split(double u, double &g, int &e)
{
    const double m = u + 0x1.8p46;
    e = __double2loint(m);                                   // round(t) + 64
    g = u - (m - 0x1.8p46);
}
__device__ __forceinline__ double wt_exp2u_join(double p, int e)
{
    return __hiloint2double(__double2hiint(p) + (e << 20), __double2loint(p));
}
// Table form: m = u + 1.5 * 2^(46 - b) carries round(2^b t) + 64 * 2^b = 2^b q + j (b = 5 or 6); 2^t = 2^(q - 64) *
// 2^(j / 2^b) * 2^(64 g) with |64 g| <= 2^-(b+1): a table of 2^b doubles in LDS (32 entries: the 64 dwords are the 64
// banks, conflict-free; 64 entries: two-way at worst) and a polynomial of degree 5 (2.2e-16) or 4 (2.4e-15; with the
// quantisation of t 5.2e-15 relative in all) - 7 or 6 double-precision FMAs per weight instead of 12 at the same
// instruction count, which the chip returns as clock (section 3.6 of DESIGN.md).  tools/make_exp2_64.py prints both.
// Round 5, later: 512 entries (4 KiB of LDS, computed by the workgroup itself with wt_exp2_64 instead of read from
// constant memory) and degree 3 - |64 g| <= 2^-10; 4.0e-15 relative in all, one FMA per weight fewer again.
#ifndef WT_BIL64_TABLE
#define WT_BIL64_TABLE 512
#endif
#define WT_EXP2T_C WT_PASTE(WT_EXP2T_C, WT_TU_NAME)
#define WT_EXP2T_T WT_PASTE(WT_EXP2T_T, WT_TU_NAME)
#if WT_BIL64_TABLE == 32
#define WT_EXP2T_DEG 5
#define WT_EXP2T_BITS 5
__constant__ double WT_EXP2T_C[6] = {0x1.0000000000000p+0, 0x1.62e42fefa39efp+5, 0x1.ebfbdff7feebap+9, 0x1.c6b08d70380ddp+13, 0x1.3b2b301f1eb9cp+17,
                                     0x1.5d885e6ef14a6p+20};
__constant__ double WT_EXP2T_T[32] = {0x1.0000000000000p-64, 0x1.059b0d3158574p-64, 0x1.0b5586cf9890fp-64, 0x1.11301d0125b51p-64, 0x1.172b83c7d517bp-64, 0x1.1d4873168b9aap-64, 0x1.2387a6e756238p-64, 0x1.29e9df51fdee1p-64, 0x1.306fe0a31b715p-64, 0x1.371a7373aa9cbp-64, 0x1.3dea64c123422p-64, 0x1.44e086061892dp-64, 0x1.4bfdad5362a27p-64, 0x1.5342b569d4f82p-64, 0x1.5ab07dd485429p-64, 0x1.6247eb03a5585p-64, 0x1.6a09e667f3bcdp-64, 0x1.71f75e8ec5f74p-64, 0x1.7a11473eb0187p-64, 0x1.82589994cce13p-64, 0x1.8ace5422aa0dbp-64, 0x1.93737b0cdc5e5p-64, 0x1.9c49182a3f090p-64, 0x1.a5503b23e255dp-64, 0x1.ae89f995ad3adp-64, 0x1.b7f76f2fb5e47p-64, 0x1.c199bdd85529cp-64, 0x1.cb720dcef9069p-64, 0x1.d5818dcfba487p-64, 0x1.dfc97337b9b5fp-64, 0x1.ea4afa2a490dap-64, 0x1.f50765b6e4540p-64};
#elif WT_BIL64_TABLE == 64
#define WT_EXP2T_DEG 4
#define WT_EXP2T_BITS 6
__constant__ double WT_EXP2T_C[5] = {0x1.0000000000000p+0, 0x1.62e42fefa0352p+5, 0x1.ebfbdff82ac52p+9, 0x1.c6b0c40d8c4e9p+13, 0x1.3b2ad0385b409p+17};
__constant__ double WT_EXP2T_T[64] = {0x1.0000000000000p-64, 0x1.02c9a3e778061p-64, 0x1.059b0d3158574p-64, 0x1.0874518759bc8p-64,
 0x1.0b5586cf9890fp-64, 0x1.0e3ec32d3d1a2p-64, 0x1.11301d0125b51p-64, 0x1.1429aaea92de0p-64,
 0x1.172b83c7d517bp-64, 0x1.1a35beb6fcb75p-64, 0x1.1d4873168b9aap-64, 0x1.2063b88628cd6p-64,
 0x1.2387a6e756238p-64, 0x1.26b4565e27cddp-64, 0x1.29e9df51fdee1p-64, 0x1.2d285a6e4030bp-64,
 0x1.306fe0a31b715p-64, 0x1.33c08b26416ffp-64, 0x1.371a7373aa9cbp-64, 0x1.3a7db34e59ff7p-64,
 0x1.3dea64c123422p-64, 0x1.4160a21f72e2ap-64, 0x1.44e086061892dp-64, 0x1.486a2b5c13cd0p-64,
 0x1.4bfdad5362a27p-64, 0x1.4f9b2769d2ca7p-64, 0x1.5342b569d4f82p-64, 0x1.56f4736b527dap-64,
 0x1.5ab07dd485429p-64, 0x1.5e76f15ad2148p-64, 0x1.6247eb03a5585p-64, 0x1.6623882552225p-64,
 0x1.6a09e667f3bcdp-64, 0x1.6dfb23c651a2fp-64, 0x1.71f75e8ec5f74p-64, 0x1.75feb564267c9p-64,
 0x1.7a11473eb0187p-64, 0x1.7e2f336cf4e62p-64, 0x1.82589994cce13p-64, 0x1.868d99b4492edp-64,
 0x1.8ace5422aa0dbp-64, 0x1.8f1ae99157736p-64, 0x1.93737b0cdc5e5p-64, 0x1.97d829fde4e50p-64,
 0x1.9c49182a3f090p-64, 0x1.a0c667b5de565p-64, 0x1.a5503b23e255dp-64, 0x1.a9e6b5579fdbfp-64,
 0x1.ae89f995ad3adp-64, 0x1.b33a2b84f15fbp-64, 0x1.b7f76f2fb5e47p-64, 0x1.bcc1e904bc1d2p-64,
 0x1.c199bdd85529cp-64, 0x1.c67f12e57d14bp-64, 0x1.cb720dcef9069p-64, 0x1.d072d4a07897cp-64,
 0x1.d5818dcfba487p-64, 0x1.da9e603db3285p-64, 0x1.dfc97337b9b5fp-64, 0x1.e502ee78b3ff6p-64,
 0x1.ea4afa2a490dap-64, 0x1.efa1bee615a27p-64, 0x1.f50765b6e4540p-64, 0x1.fa7c1819e90d8p-64};
#elif WT_BIL64_TABLE == 512
#define WT_EXP2T_DEG 3
#define WT_EXP2T_BITS 9
#define WT_EXP2T_COMPUTED 1          // no constant table: entry j = wt_exp2_64(j / 512 - 64)
__constant__ double WT_EXP2T_C[4] = {0x1.ffffffffffff6p-1, 0x1.62e42fefa39eep+5, 0x1.ebfbe13357103p+9, 0x1.c6b08e1f0e0b5p+13};
#endif
#if WT_BIL64_TABLE
__device__ __forceinline__ void wt_exp2t_split(double u, double &g, int &e)
{
    constexpr double M = WT_EXP2T_BITS == 5 ? 0x1.8p41 : (WT_EXP2T_BITS == 6 ? 0x1.8p40 : 0x1.8p37);
    static_assert(WT_EXP2T_BITS == 5 || WT_EXP2T_BITS == 6 || WT_EXP2T_BITS == 9, "1.5 * 2^(46 - bits)");
    const double m = u + M;
    e = __double2loint(m);                                   // 2^b (round-ish(t) + 64) + j
    g = u - (m - M);
}
__device__ __forceinline__ double wt_exp2t_join(double p, double tj, int e)
{
    const double w = p * tj;
    return __hiloint2double(__double2hiint(w) + ((e & ~(WT_BIL64_TABLE - 1)) << (20 - WT_EXP2T_BITS)), __double2loint(w));
}
#endif
// (scalar form: the kernels evaluate several weights in lockstep, see wt64_bilateral_march_kernel)
__device__ __forceinline__ double wt_exp2_64_from_u(double u)
{
    const double *C = WT_EXP2U_C;
    double g;
    int e;
    wt_exp2u_split(u, g, e);
    double p = C[10];
#pragma unroll
    for (int k = 9; k >= 0; --k) p = fma(p, g, C[k]);
    return wt_exp2u_join(p, e);
}
__device__ __forceinline__ double wt_exp2_64(double t)
{
    const double *C = WT_EXP2_64_C;
    t = fmax(t, -1020.0);
    const double m = t + 0x1.8p52;
    const double f = t - (m - 0x1.8p52);
    double p = C[10];
#pragma unroll
    for (int k = 9; k >= 0; --k) p = fma(p, f, C[k]);
    const int n = __double2loint(m);                         // round(t) in two's complement
    return __hiloint2double(__double2hiint(p) + (n << 20), __double2loint(p));
}
