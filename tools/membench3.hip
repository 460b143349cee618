// Which feature of the fused pass-A geometry costs the ~10 % between its memory pattern
// (tools/membench2: 5.2 TB/s) and the real kernel with the arithmetic removed (4.5 TB/s)?
// One kernel, every feature a run-time switch:
//   vx / hx   : valid pixels per strip and halo per side (lanes = 1024 pixels per 4-wave workgroup);
//               halo lanes load (clamped) but do not store
//   lag       : planes 0..4 store row r-2, r-7, r-16, r-16, r-16 (the cascade's lags) instead of r
//   warm      : rows read before the first stored row of a chunk (28 in pass A)
//   chunks    : chunks per column strip (57 in pass A at 8192^2)
// build: hipcc -O3 --offload-arch=gfx950 -o tools/membench3 tools/membench3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
struct Ptrs { float4 *p[8]; };
typedef float vf4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ntstore(float4 o, float4 *p) { vf4 v = {o.x, o.y, o.z, o.w}; __builtin_nontemporal_store(v, (vf4 *)p); }

struct Cfg { int W4, H, nstrips, vx4, hx4, S, chunks, warm, lag, nwr, D, nohaloload, nomask, rdp; };

template <int PD>
__global__ __launch_bounds__(512) void march(Ptrs in, Ptrs out, Cfg c)
{
    const int strip = blockIdx.x % c.nstrips, item = blockIdx.x / c.nstrips;
    const int q = item % c.D, chunk = item / c.D;
    const int n_q = (c.H - q + c.D - 1) / c.D;
    const int r0 = chunk * c.S, r1 = min(r0 + c.S, n_q);
    if (r0 >= r1) return;
    const int x4 = strip * c.vx4 - c.hx4 + (int)threadIdx.x;          // float4 column (may be outside)
    const bool valid = x4 >= strip * c.vx4 && x4 < (strip + 1) * c.vx4 && x4 < c.W4;
    const bool st = valid || (c.nomask && x4 >= 0 && x4 < c.W4);
    const bool ld = valid || !c.nohaloload;
    const int xl = min(max(x4, 0), c.W4 - 1);
    auto row = [&](int r) -> long { return (long)(q + c.D * min(max(r, 0), n_q - 1)) * c.W4 + xl; };
    const int t0 = r0 - c.warm;
    const float4 z = make_float4(0, 0, 0, 0);
    float4 pf[PD], pp[PD];
#pragma unroll
    for (int i = 0; i < PD; ++i) { pf[i] = ld ? in.p[0][row(t0 + i)] : z; pp[i] = (c.rdp && valid) ? in.p[1][row(t0 + i)] : z; }
    const int lags[5] = {2, 7, 16, 16, 16};
    for (int t = t0; t < r1 + (c.lag ? 16 : 0); t += PD) {
#pragma unroll
        for (int k = 0; k < PD; ++k) {
            const int tt = t + k;
            float4 cur = pf[k];
            cur.x += pp[k].x;
            pf[k] = ld ? in.p[0][row(tt + PD)] : z;
            if (c.rdp && valid) pp[k] = in.p[1][row(tt + PD)];
            for (int w = 0; w < c.nwr; ++w) {
                const int r = tt - (c.lag ? lags[w] : 0);
                if (st && r >= r0 && r < r1) {
                    const long o = (long)(q + c.D * r) * c.W4 + x4;
                    if (w < 3) ntstore(cur, &out.p[w][o]); else out.p[w][o] = cur;
                }
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

int main()
{
    const int side = 8192;
    const long n4 = (long)side * side / 4;
    const size_t bytes = (size_t)n4 * 16;
    Ptrs in{}, out{};
    CK(hipMalloc(&in.p[0], bytes)); CK(hipMemset(in.p[0], 0, bytes));
    for (int i = 0; i < 5; ++i) CK(hipMalloc(&out.p[i], bytes + 4352 * i));
    CK(hipMalloc(&in.p[1], bytes)); CK(hipMemset(in.p[1], 0, bytes));
    auto run = [&](const char *name, int nw, int D, int vx, int hx, int wgs, int warm, int lag, int nohaloload, int nomask, int rdp) {
        Cfg c{};
        c.W4 = side / 4; c.H = side; c.vx4 = vx / 4; c.hx4 = hx / 4; c.D = D;
        c.nstrips = (side + vx - 1) / vx;
        c.chunks = wgs / (c.nstrips * D) > 0 ? wgs / (c.nstrips * D) : 1;
        const int n_q = (side + D - 1) / D;
        c.S = (n_q + c.chunks - 1) / c.chunks;
        c.warm = warm; c.lag = lag; c.nwr = 5; c.nohaloload = nohaloload; c.nomask = nomask; c.rdp = rdp;
        const int grid = c.nstrips * D * c.chunks;
        double ms = timeit([&] { hipLaunchKernelGGL((march<4>), dim3(grid), dim3(nw * 64), 0, 0, in, out, c); });
        const double planes = 6.0 + rdp;
        printf("%-46s NW %d D %d vx %4d hx %3d strips %2d WGs %4d S %3d warm %2d lag %d : %.4f ms  (%.0f planes: %5.0f GB/s)\n",
               name, nw, D, vx, hx, c.nstrips, grid, c.S, warm, lag, ms, planes, planes * bytes / ms / 1e6);
    };
    for (int rep = 0; rep < 2; ++rep) {
        run("A: pattern only (8 x 1024, no halo)", 4, 1, 1024, 0, 512, 0, 0, 0, 0, 0);
        run("A: halo 32, strips 960, masked stores", 4, 1, 960, 32, 512, 0, 0, 0, 0, 0);
        run("A: same, halo lanes do not load", 4, 1, 960, 32, 512, 0, 0, 1, 0, 0);
        run("A: same, halo lanes load AND store (overlap)", 4, 1, 960, 32, 512, 0, 0, 0, 1, 0);
        run("A: 9 strips of 960 without any halo lanes", 4, 1, 960, 0, 512, 0, 0, 0, 0, 0);
        run("A: real geometry 928/32 + lags + warm-up", 4, 1, 928, 32, 512, 28, 1, 0, 0, 0);
        run("B: pattern only D=8 (8 x 1024 x 4 waves)", 4, 8, 1024, 0, 512, 0, 0, 0, 0, 1);
        run("B: pattern only D=8, 8-wave WGs (4 x 2048)", 8, 8, 2048, 0, 256, 0, 0, 0, 0, 1);
        run("B: 8 waves, strips 1664 halo 128 masked", 8, 8, 1664, 128, 256, 0, 0, 0, 0, 1);
        run("B: same, halo lanes do not load", 8, 8, 1664, 128, 256, 0, 0, 1, 0, 1);
        run("B: real geometry + lags + warm-up 14", 8, 8, 1664, 128, 256, 14, 1, 0, 0, 1);
        run("B: 4 waves, strips 768 halo 128", 4, 8, 768, 128, 512, 14, 1, 0, 0, 1);
    }
    return 0;
}
