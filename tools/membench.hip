// HBM streaming microbenchmark: ceilings for the read/write mixes of the a-trous kernels.
// build: hipcc -O3 --offload-arch=gfx950 -o tools/membench tools/membench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef float vf4 __attribute__((ext_vector_type(4)));
struct Ptrs { float4 *p[12]; };
__device__ __forceinline__ float4 ntload(const float4 *p) { vf4 v = __builtin_nontemporal_load((const vf4 *)p); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void ntstore(float4 o, float4 *p) { vf4 v = {o.x, o.y, o.z, o.w}; __builtin_nontemporal_store(v, (vf4 *)p); }

template <int NR, int NWR, int UNROLL, int NT>
__global__ __launch_bounds__(256) void stream_kernel(Ptrs in, Ptrs out, long n4)
{
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride * UNROLL) {
        float4 acc[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            acc[u] = make_float4(0, 0, 0, 0);
            const long j = i + u * stride;
            if (j < n4) {
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    float4 v = (NT & 1) ? ntload(&in.p[r][j]) : in.p[r][j];
                    acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long j = i + u * stride;
            if (j < n4) {
#pragma unroll
                for (int w = 0; w < NWR; ++w) {
                    float4 o = acc[u]; o.x += w;
                    if (NT & 2) ntstore(o, &out.p[w][j]); else out.p[w][j] = o;
                }
            }
        }
    }
}

// block-contiguous variant: each block streams a contiguous tile of `tile` float4
template <int NR, int NWR>
__global__ __launch_bounds__(256) void tile_kernel(Ptrs in, Ptrs out, long n4, long tile)
{
    const long b0 = (long)blockIdx.x * tile;
    const long b1 = min(b0 + tile, n4);
    for (long i = b0 + threadIdx.x; i < b1; i += 256) {
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NR; ++r) { float4 v = in.p[r][i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
#pragma unroll
        for (int w = 0; w < NWR; ++w) { float4 o = acc; o.x += w; out.p[w][i] = o; }
    }
}

// Row-march pattern of the fused a-trous pass, without the arithmetic: a 256-thread workgroup
// owns a 1024-pixel column strip and walks down `rows` consecutive image rows, reading one row
// segment (16 B per lane) and writing it to 4 planes, with PD rows of software prefetch.
template <int PD>
__global__ __launch_bounds__(256) void march_kernel(Ptrs in, Ptrs out, int W4, int rows, int nstrips)
{
    const int strip = blockIdx.x % nstrips, chunk = blockIdx.x / nstrips;
    const long base = (long)chunk * rows * W4 + (long)strip * 256 + threadIdx.x;
    float4 pf[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) pf[i] = in.p[0][base + (long)min(i, rows - 1) * W4];
    for (int r = 0; r < rows; r += PD) {
#pragma unroll
        for (int k = 0; k < PD; ++k) {
            const int rr = r + k;
            if (rr >= rows) break;
            const float4 cur = pf[k];
            pf[k] = in.p[0][base + (long)min(rr + PD, rows - 1) * W4];
            const long o = base + (long)rr * W4;
            out.p[0][o] = cur; out.p[1][o] = cur; out.p[2][o] = cur; out.p[3][o] = cur;
        }
    }
}

// the same march with 8 bytes per lane (float2): half the registers per lane in the real kernel,
// i.e. twice the resident waves - does the narrower access still stream?
template <int PD>
__global__ __launch_bounds__(256) void march2_kernel(Ptrs in, Ptrs out, int W2, int rows, int nstrips)
{
    const int strip = blockIdx.x % nstrips, chunk = blockIdx.x / nstrips;
    const long base = (long)chunk * rows * W2 + (long)strip * 256 + threadIdx.x;
    const float2 *src = (const float2 *)in.p[0];
    float2 pf[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) pf[i] = src[base + (long)min(i, rows - 1) * W2];
    for (int r = 0; r < rows; r += PD) {
#pragma unroll
        for (int k = 0; k < PD; ++k) {
            const int rr = r + k;
            if (rr >= rows) break;
            const float2 cur = pf[k];
            pf[k] = src[base + (long)min(rr + PD, rows - 1) * W2];
            const long o = base + (long)rr * W2;
            ((float2 *)out.p[0])[o] = cur; ((float2 *)out.p[1])[o] = cur;
            ((float2 *)out.p[2])[o] = cur; ((float2 *)out.p[3])[o] = cur;
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
    const long side = argc > 1 ? atol(argv[1]) : 8192;
    const long n4 = side * side / 4;
    const size_t bytes = (size_t)n4 * 16;
    Ptrs in{}, out{};
    for (int i = 0; i < 8; ++i) { CK(hipMalloc(&in.p[i], bytes)); CK(hipMemset(in.p[i], 0, bytes)); }
    for (int i = 0; i < 5; ++i) { CK(hipMalloc(&out.p[i], bytes)); }
    printf("plane %ld x %ld (%.0f MiB)\n", side, side, bytes / 1048576.0);
#define RUN(NR, NWR, UN, NT, GRID)                                                              \
    {                                                                                           \
        double ms = timeit([&] { hipLaunchKernelGGL((stream_kernel<NR, NWR, UN, NT>), dim3(GRID), dim3(256), 0, 0, in, out, n4); }); \
        printf("R%d W%d unroll%d nt%d grid%-6d : %.4f ms  %.0f GB/s\n", NR, NWR, UN, NT, GRID, ms, (NR + NWR) * (double)bytes / ms / 1e6); \
    }
    RUN(7, 1, 1, 0, 2048) RUN(7, 1, 1, 1, 2048) RUN(7, 1, 1, 2, 2048) RUN(7, 1, 1, 3, 2048)
    RUN(7, 1, 1, 1, 8192) RUN(7, 1, 1, 1, 16384) RUN(7, 1, 1, 1, 65536) RUN(7, 1, 1, 3, 16384) RUN(7, 1, 1, 3, 65536) RUN(7, 1, 1, 3, 262144)
    RUN(7, 1, 1, 1, 512) RUN(7, 1, 1, 1, 1024) RUN(7, 1, 1, 3, 1024)
    RUN(1, 4, 1, 0, 2048) RUN(1, 4, 1, 1, 2048) RUN(1, 4, 1, 0, 16384) RUN(1, 4, 1, 0, 65536) RUN(1, 4, 1, 1, 65536)
    RUN(1, 1, 1, 0, 65536) RUN(1, 1, 1, 1, 65536) RUN(1, 1, 1, 3, 65536) RUN(1, 1, 1, 0, 262144)
    {
        const int W4 = side / 4, nstrips = W4 / 256;
        for (int rows : {2048, 1024, 512, 256, 128, 64, 32}) {
            const int chunks = side / rows, grid = chunks * nstrips;
#define MARCH(PD)                                                                                  \
            {                                                                                      \
                double ms = timeit([&] { hipLaunchKernelGGL((march_kernel<PD>), dim3(grid), dim3(256), 0, 0, in, out, W4, rows, nstrips); }); \
                printf("march R1W4 rows/WG %4d  WGs %5d  PD%d : %.4f ms  %.0f GB/s\n", rows, grid, PD, ms, 5.0 * bytes / ms / 1e6); \
            }
            MARCH(1) MARCH(4) MARCH(8)
        }
    }
    {
        const int W2 = side / 2, nstrips = W2 / 256;
        for (int rows : {512, 256, 128, 64}) {
            const int chunks = side / rows, grid = chunks * nstrips;
            double ms = timeit([&] { hipLaunchKernelGGL((march2_kernel<1>), dim3(grid), dim3(256), 0, 0, in, out, W2, rows, nstrips); });
            printf("march2 (8 B/lane) R1W4 rows/WG %4d  WGs %5d  PD1 : %.4f ms  %.0f GB/s\n", rows, grid, ms, 5.0 * bytes / ms / 1e6);
            ms = timeit([&] { hipLaunchKernelGGL((march2_kernel<4>), dim3(grid), dim3(256), 0, 0, in, out, W2, rows, nstrips); });
            printf("march2 (8 B/lane) R1W4 rows/WG %4d  WGs %5d  PD4 : %.4f ms  %.0f GB/s\n", rows, grid, ms, 5.0 * bytes / ms / 1e6);
        }
    }
    for (long tile : {4096L}) {
        long grid = (n4 + tile - 1) / tile;
        double ms = timeit([&] { hipLaunchKernelGGL((tile_kernel<7, 1>), dim3(grid), dim3(256), 0, 0, in, out, n4, tile); });
        printf("tile R7 W1 tile%ld grid%ld : %.4f ms %.0f GB/s\n", tile, grid, ms, 8 * (double)bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL((tile_kernel<1, 4>), dim3(grid), dim3(256), 0, 0, in, out, n4, tile); });
        printf("tile R1 W4 tile%ld grid%ld : %.4f ms %.0f GB/s\n", tile, grid, ms, 5 * (double)bytes / ms / 1e6);
    }
    return 0;
}
