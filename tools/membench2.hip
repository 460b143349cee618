// Front-shape experiments for the fused a-trous passes (round 2): the same per-workgroup march
// (one 16-byte row load, NWR 16-byte row stores per step, PD rows of prefetch, 4-wave workgroups
// 1024 pixels wide) with different assignments of rows to workgroups.  Question: how much of the
// gap between the march (~5.0 TB/s) and a flat stream (~5.9 TB/s) is DRAM page locality, i.e.
// does it close when the workgroups that run at the same time write ADJACENT rows?
//   mode "chunk":  workgroup (strip, j) walks rows j*S .. j*S+S-1            (pass A today)
//   mode "phase":  workgroup (strip, q, c) walks rows q + D*(c*S + k)        (pass B: D = 8)
// build: hipcc -O3 --offload-arch=gfx950 -o tools/membench2 tools/membench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
struct Ptrs { float4 *p[8]; };
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ntstore(float4 o, float4 *p) { vf4 v = {o.x, o.y, o.z, o.w}; __builtin_nontemporal_store(v, (vf4 *)p); }

// NWR planes written, NRD row loads per step (NRD = 2: the second load reads the row D/2 above -
// the "every phase loads both parities" variant), NT: nontemporal stores to planes 0..2
template <int PD, int NWR, int NRD, int NT, int PERM = 0>
__global__ __launch_bounds__(256) void march(Ptrs in, Ptrs out, int W4, int H, int nstrips, int D, int S, int chunks)
{
    const int strip = blockIdx.x % nstrips;
    const int item = blockIdx.x / nstrips;
    const int q = item % D, c = item / D;
    if (c >= chunks) return;
    const int n_q = (H - q + D - 1) / D;
    const int r0 = c * S, r1 = min(r0 + S, n_q);
    if (r0 >= r1) return;
    const long col = (long)strip * 256 + threadIdx.x;
    // PERM: chunk-interleaved row layout - logical row y = j*S + k lives at physical row k*chunks + j,
    // so the rows that the `chunks` workgroups of a column strip touch in the same step are adjacent
    auto phys = [&](int y) -> long { return PERM ? (long)(y % S) * chunks + y / S : (long)y; };
    auto row = [&](int r) -> long { return phys(q + D * min(r, r1 - 1)) * W4 + col; };
    float4 pf[PD], pg[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) { pf[i] = in.p[0][row(r0 + i)]; if (NRD > 1) pg[i] = in.p[0][max(row(r0 + i) - (long)(D / 2) * W4, col)]; }
    for (int r = r0; r < r1; r += PD) {
#pragma unroll
        for (int k = 0; k < PD; ++k) {
            const int rr = r + k;
            if (rr >= r1) break;
            float4 cur = pf[k];
            pf[k] = in.p[0][row(rr + PD)];
            if (NRD > 1) { cur.x += pg[k].x; pg[k] = in.p[0][max(row(rr + PD) - (long)(D / 2) * W4, col)]; }
            const long o = phys(q + D * rr) * W4 + col;
#pragma unroll
            for (int w = 0; w < NWR; ++w) {
                if (NT && w < 3) ntstore(cur, &out.p[w][o]); else out.p[w][o] = cur;
            }
        }
    }
}

template <typename F>
static double timeit(F f, int reps = 20)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int side = argc > 1 ? atoi(argv[1]) : 8192;
    const long n4 = (long)side * side / 4;
    const size_t bytes = (size_t)n4 * 16;
    Ptrs in{}, out{};
    CK(hipMalloc(&in.p[0], bytes)); CK(hipMemset(in.p[0], 0, bytes));
    for (int i = 0; i < 5; ++i) CK(hipMalloc(&out.p[i], bytes + 65536 * i));
    const int W4 = side / 4, nstrips = W4 / 256;
    printf("plane %d x %d; per step 1 row load + NWR row stores; GB/s counts (1 + NWR) planes\n", side, side);
    for (int rep = 0; rep < 2; ++rep)
    for (int D : {1, 2, 4, 8, 16}) {
        for (int wgs : {512, 256, 1024}) {
            const int chunks = wgs / (nstrips * D) > 0 ? wgs / (nstrips * D) : 1;
            const int n_q = (side + D - 1) / D;
            const int S = (n_q + chunks - 1) / chunks;
            const int grid = nstrips * D * chunks;
#define GO(NWR, NRD, NT)                                                                                \
            {                                                                                               \
                double ms = timeit([&] { hipLaunchKernelGGL((march<4, NWR, NRD, NT>), dim3(grid), dim3(256), 0, 0, in, out, W4, side, nstrips, D, S, chunks); }); \
                printf("D %2d  WGs %4d  S %4d  W%d R%d nt%d : %.4f ms  %5.0f GB/s\n", D, grid, S, NWR, NRD, NT, ms, (1.0 + NWR) * bytes / ms / 1e6); \
            }
            GO(5, 1, 1) GO(5, 1, 0)
            if (D == 1 && side % (chunks * 1) == 0 && S * chunks == side) {
                double ms = timeit([&] { hipLaunchKernelGGL((march<4, 5, 1, 1, 1>), dim3(grid), dim3(256), 0, 0, in, out, W4, side, nstrips, D, S, chunks); });
                printf("D %2d  WGs %4d  S %4d  W5 R1 nt1 PERMUTED rows : %.4f ms  %5.0f GB/s\n", D, grid, S, ms, 6.0 * bytes / ms / 1e6);
            }
            if (wgs == 512) { GO(4, 1, 1) if (D > 1) GO(5, 2, 1) }
        }
    }
    return 0;
}
