/* Plain-C client of the C ABI (include/watroo_hip.h): no Python, no C++, no HIP headers.
 *
 *   gcc -O2 -Iinclude examples/abi_demo.c -o examples/abi_demo -Lwavelets_amd -lwatroo_hip \
 *       -Wl,-rpath,$PWD/wavelets_amd -lm
 *   ./examples/abi_demo [H W level]
 *
 * Runs the hot path on a synthetic image - upload, wt_decompose_sum (B3spline), exact MAD noise
 * estimate, soft-threshold of the first two planes, plane sum - and checks the two size-
 * independent properties the transform has: sum of planes == input (to rounding) and the
 * carried sum == wt_plane_sum bit for bit.  Exit code 0 = ok.  tests/test_gpu_parity.py builds
 * and runs it. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "watroo_hip.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        if ((call) != 0) {                                                           \
            fprintf(stderr, "%s failed: %s\n", #call, wt_last_error());             \
            return 2;                                                                \
        }                                                                            \
    } while (0)

int main(int argc, char **argv)
{
    const long H = argc > 1 ? atol(argv[1]) : 600, W = argc > 2 ? atol(argv[2]) : 900;
    const int level = argc > 3 ? atoi(argv[3]) : 6;
    float *img = malloc(sizeof(float) * H * W), *rec = malloc(sizeof(float) * H * W),
          *rec2 = malloc(sizeof(float) * H * W);
    unsigned s = 12345u;
    double amax = 0.0;
    for (long i = 0; i < H * W; ++i) {        /* LCG noise in [-1, 1) plus a ramp */
        s = s * 1664525u + 1013904223u;
        img[i] = (float)((s >> 8) / 8388608.0 - 1.0) + (float)(i % W) / (float)W;
        if (fabs(img[i]) > amax) amax = fabs(img[i]);
    }
    int ndev = 0;
    CHECK(wt_device_count(&ndev));
    if (ndev < 1) { fprintf(stderr, "no HIP device: %s\n", wt_last_error()); return 3; }
    wt_ctx *ctx = NULL;
    wt_plan *plan = NULL;
    CHECK(wt_ctx_create(0, &ctx));
    CHECK(wt_plan_create(ctx, H, W, WT_B3SPLINE, level, &plan));
    CHECK(wt_upload(plan, WT_PLANE_INPUT, img, W));
    CHECK(wt_decompose_sum(plan, WT_PLANE_INPUT, level, WT_PLANE_OUT, 1 /* fused passes */));
    CHECK(wt_download(plan, WT_PLANE_OUT, rec, W));
    double worst = 0.0;
    for (long i = 0; i < H * W; ++i) worst = fmax(worst, fabs((double)rec[i] - img[i]));
    printf("sum of planes - input: max %.3e (|input| max %.3f)\n", worst, amax);
    if (worst > 1e-5 * amax) return 1;
    CHECK(wt_plane_sum(plan, 0, level + 1, WT_PLANE_SCRATCH(2)));
    CHECK(wt_download(plan, WT_PLANE_SCRATCH(2), rec2, W));
    if (memcmp(rec, rec2, sizeof(float) * H * W) != 0) { fprintf(stderr, "carried sum != plane sum\n"); return 1; }
    float med = 0.f;
    CHECK(wt_abs_median(plan, 0, &med));
    const double sigma_e0 = 8.907e-01, sigma_e1 = 2.0072e-01;     /* B3spline 2-D table */
    const double noise = (double)med / 0.6745 / sigma_e0;
    CHECK(wt_denoise(plan, 0, 5.0 * noise * sigma_e0, 1.0, 1, WT_PLANE_NONE));
    CHECK(wt_denoise(plan, 1, 3.0 * noise * sigma_e1, 1.0, 1, WT_PLANE_NONE));
    CHECK(wt_plane_sum(plan, 0, level + 1, WT_PLANE_OUT));
    CHECK(wt_download(plan, WT_PLANE_OUT, rec, W));
    double e_in = 0.0, e_out = 0.0;
    for (long i = 0; i < H * W; ++i) { e_in += (double)img[i] * img[i]; e_out += (double)rec[i] * rec[i]; }
    printf("noise estimate %.5f; energy in %.4e -> denoised %.4e\n", noise, e_in, e_out);
    if (!(noise > 0.0) || !(e_out < e_in)) return 1;
    CHECK(wt_plan_destroy(plan));
    /* the float64 engine: the same image with an offset float32 cannot hold, in double precision */
    {
        double *img64 = malloc(sizeof(double) * H * W), *rec64 = malloc(sizeof(double) * H * W);
        for (long i = 0; i < H * W; ++i) img64[i] = (double)img[i] + 1.0e7;
        const double b3[5] = {1.0 / 16, 4.0 / 16, 6.0 / 16, 4.0 / 16, 1.0 / 16};
        wt_plan64 *p64 = NULL;
        CHECK(wt64_plan_create(ctx, H, W, level, b3, 5, &p64));
        CHECK(wt64_upload(p64, WT_PLANE_INPUT, img64, W));
        CHECK(wt64_decompose(p64, WT_PLANE_INPUT, level, 0));
        CHECK(wt64_plane_sum(p64, 0, level + 1, WT_PLANE_OUT));
        CHECK(wt64_download(p64, WT_PLANE_OUT, rec64, W));
        double worst64 = 0.0, med64 = 0.0;
        for (long i = 0; i < H * W; ++i) worst64 = fmax(worst64, fabs(rec64[i] - img64[i]));
        CHECK(wt64_abs_median(p64, 0, &med64));
        printf("float64: sum of planes - input: max %.3e at offset 1e7; median |w_0| %.6f (float32 engine %.6f)\n",
               worst64, med64, (double)med);
        if (worst64 > 1e-8 || fabs(med64 - (double)med) > 1e-5 * fabs((double)med)) return 1;
        CHECK(wt64_plan_destroy(p64));
        free(img64); free(rec64);
    }
    CHECK(wt_ctx_destroy(ctx));
    free(img); free(rec); free(rec2);
    printf("abi_demo: OK\n");
    return 0;
}
