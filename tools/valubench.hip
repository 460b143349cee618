// VALU issue-cost microbenchmark for the bilateral kernel's instruction mix (gfx950).
//   hipcc -O3 --offload-arch=gfx950 -o tools/valubench tools/valubench.hip && tools/valubench
// Every test is a workgroup-free kernel: W waves per SIMD (grid = 256 CUs x 4 SIMDs x W waves), each wave runs
// ITERS iterations of an unrolled body of inline-asm instructions on its own registers.  Reported: nanoseconds
// and core cycles (s_memtime ticks at 100 MHz are no use: the clock is derived from v_fma_f32 = 4 cycles at one
// wave per SIMD... so the table is RELATIVE to v_fma_f32 = 4.00) per instruction per SIMD.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

static int ITERS = 2000, LAUNCHES = 5;   // argv[1], argv[2]

#define REP4(x) x x x x
#define REP8(x) REP4(x) REP4(x)

// 8 independent chains per wave, so a single wave never waits on its own result
template <int T>
__global__ __launch_bounds__(64) void wt_valubench(float *out, float seed, int iters)
{
    f2 a0 = {seed, seed + 1}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    f2 b = {seed * 0.5f, seed * 0.25f}, c = {1e-3f, 2e-3f};
    f2 e0 = a0, e1 = a1, e2 = a2, e3 = a3;
    for (int it = 0; it < iters; ++it) {
        if constexpr (T == 0) {          // v_fma_f32 x 16
            asm volatile(
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x)
                : "v"(b.x), "v"(c.x));
        } else if constexpr (T == 1) {   // v_pk_fma_f32 x 16
            asm volatile(
                "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(b), "v"(c));
        } else if constexpr (T == 2) {   // v_pk_add_f32 x 16
            asm volatile(
                "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                "v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(c));
        } else if constexpr (T == 3) {   // v_pk_mul_f32 x 16
            asm volatile(
                "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                "v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(b));
        } else if constexpr (T == 4) {   // v_exp_f32 x 16
            asm volatile(
                "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                : "+v"(a0.x), "+v"(a1.x), "+v"(a2.x), "+v"(a3.x), "+v"(a4.x), "+v"(a5.x), "+v"(a6.x), "+v"(a7.x));
        } else if constexpr (T == 5) {
            // the tap of wt_bilateral2_kernel as the product source writes it (the compiler packs it), 8 taps per iteration;
            // the empty asm makes the operands opaque so that nothing is hoisted out of the loop
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(b));
            const f2 lk = {-4.f, -4.f};
            const f2 ts[8] = {a0, a1, a2, a3, a0 + c, a1 + c, a2 + c, a3 + c};
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const f2 diff = a4 - ts[t];
                const f2 ex = __builtin_elementwise_fma(diff * diff, b, lk);
                const f2 w = {__builtin_amdgcn_exp2f(ex.x), __builtin_amdgcn_exp2f(ex.y)};
                a5 += w;
                a6 = __builtin_elementwise_fma(ts[t], w, a6);
            }
        } else if constexpr (T == 7) {
            // the same 8 taps with nothing packed: one instruction per pixel and operation
            asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(b));
            const f2 ts[8] = {a0, a1, a2, a3, a0 + c, a1 + c, a2 + c, a3 + c};
            float nx = a5.x, ny = a5.y, cx = a6.x, cy = a6.y;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                float dx, dy, wx, wy;
                asm volatile("v_sub_f32 %0, %2, %4\n v_sub_f32 %1, %3, %5\n v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n"
                             "v_fma_f32 %0, %0, %6, -4.0\n v_fma_f32 %1, %1, %7, -4.0\n"
                             : "=&v"(dx), "=&v"(dy)
                             : "v"(a4.x), "v"(a4.y), "v"(ts[t].x), "v"(ts[t].y), "v"(b.x), "v"(b.y));
                wx = __builtin_amdgcn_exp2f(dx);
                wy = __builtin_amdgcn_exp2f(dy);
                asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %5\n v_fma_f32 %2, %6, %4, %2\n v_fma_f32 %3, %7, %5, %3\n"
                             : "+v"(nx), "+v"(ny), "+v"(cx), "+v"(cy)
                             : "v"(wx), "v"(wy), "v"(ts[t].x), "v"(ts[t].y));
            }
            a5 = (f2){nx, ny};
            a6 = (f2){cx, cy};
        } else if constexpr (T == 8) {   // v_pk_fma_f32 with an SGPR-pair addend (the log2 tap weight), 16 of them
            asm volatile(
                "v_pk_fma_f32 %0, %0, %8, s[2:3] op_sel_hi:[1,1,0]\n v_pk_fma_f32 %1, %1, %8, s[2:3] op_sel_hi:[1,1,0]\n"
                "v_pk_fma_f32 %2, %2, %8, s[2:3] op_sel_hi:[1,1,0]\n v_pk_fma_f32 %3, %3, %8, s[2:3] op_sel_hi:[1,1,0]\n"
                "v_pk_fma_f32 %4, %4, %8, s[2:3] op_sel_hi:[1,1,0]\n v_pk_fma_f32 %5, %5, %8, s[2:3] op_sel_hi:[1,1,0]\n"
                "v_pk_fma_f32 %6, %6, %8, s[2:3] op_sel_hi:[1,1,0]\n v_pk_fma_f32 %7, %7, %8, s[2:3] op_sel_hi:[1,1,0]\n"
                "v_pk_fma_f32 %0, %0, %8, s[2:3] op_sel_hi:[1,1,0]\n v_pk_fma_f32 %1, %1, %8, s[2:3] op_sel_hi:[1,1,0]\n"
                "v_pk_fma_f32 %2, %2, %8, s[2:3] op_sel_hi:[1,1,0]\n v_pk_fma_f32 %3, %3, %8, s[2:3] op_sel_hi:[1,1,0]\n"
                "v_pk_fma_f32 %4, %4, %8, s[2:3] op_sel_hi:[1,1,0]\n v_pk_fma_f32 %5, %5, %8, s[2:3] op_sel_hi:[1,1,0]\n"
                "v_pk_fma_f32 %6, %6, %8, s[2:3] op_sel_hi:[1,1,0]\n v_pk_fma_f32 %7, %7, %8, s[2:3] op_sel_hi:[1,1,0]\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(b));
        } else if constexpr (T == 9) {   // v_fma_f64 x 16 (the float64 march's currency)
            asm volatile(
                "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                : "v"(b), "v"(c));
        }
    }
    f2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + e0 + e1 + e2 + e3 + b + c;
    if (s.x == 12345.678f) out[threadIdx.x] = s.x + s.y;   // keeps the chains alive
}

static const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_exp_f32", "8 taps (packed, as the kernel writes them)",
                              "-", "8 taps (nothing packed)", "v_pk_fma_f32 sgpr addend", "v_fma_f64"};
static const int ninst[] = {16, 16, 16, 16, 16, 8, 1, 8, 16, 16};   // (tap rows: per TAP of two pixels)

template <int T>
static double run(int waves_per_simd, float *out)
{
    const int grid = 256 * 4 * waves_per_simd;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(wt_valubench<T>, dim3(grid), dim3(64), 0, 0, out, 1.0f, ITERS);
    hipEventRecord(e0, 0);
    for (int i = 0; i < LAUNCHES; ++i) hipLaunchKernelGGL(wt_valubench<T>, dim3(grid), dim3(64), 0, 0, out, 1.0f, ITERS);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / LAUNCHES * 1e6 / ((double)ITERS * ninst[T] * waves_per_simd);   // ns per instruction per SIMD
}

int main(int argc, char **argv)
{
    if (argc > 1) ITERS = atoi(argv[1]);
    if (argc > 2) LAUNCHES = atoi(argv[2]);
    float *out;
    hipMalloc(&out, 4096);
    printf("%-52s %10s %10s %10s %10s   (ns per instruction per SIMD; in brackets: cycles if v_fma_f32 at that occupancy = 4)\n", "test", "1 w/SIMD", "2 w/SIMD",
           "4 w/SIMD", "5 w/SIMD");
    double base[4] = {0, 0, 0, 0};
    const int ws[4] = {1, 2, 4, 5};
#define ROW(T)                                                                                      \
    {                                                                                               \
        printf("%-52s", names[T]);                                                                  \
        for (int i = 0; i < 4; ++i) {                                                               \
            const double ns = run<T>(ws[i], out);                                                   \
            if (T == 0) base[i] = ns;                                                               \
            printf(" %5.2f(%4.1f)", ns, 4.0 * ns / base[i]);                                        \
        }                                                                                           \
        printf("\n");                                                                               \
        fflush(stdout);                                                                             \
    }
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(8) ROW(9) ROW(5) ROW(7)
    printf("tap rows: per TAP of two pixels (7 instructions packed, 12 un-packed)\n");
    return 0;
}
