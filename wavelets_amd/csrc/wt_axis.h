// Tiled 1-D filters with RUN-TIME taps along one axis of a plane (round 5): the kernels behind wt_axis_filter /
// wt64_axis_filter, which the host mirror uses axis by axis for scaling functions the tuned kernels do not
// take - an even number of taps, more than 15 (watroo/wavelets.py:152-197 puts no limit on coefficients_1d) -
// in the standard and the recursive algorithm, on signals, images and cubes.  Until now these ran on the
// one-sample-per-thread tap-list operator (wt_taps_kernel): K global loads per sample per axis, 0.28 ms per
// scale for 17 taps at 2048^2.  Here every input sample is fetched from memory once per axis:
//   * along x a workgroup stages a row segment and its halo in LDS (border rule applied while loading) and
//     every lane reads its taps from there - conflict-free, lanes own pixels 256 apart;
//   * along y / z a workgroup stages the rows of one chunk of a polyphase chain (64 columns wide, its K - 1
//     halo rows included) in LDS with all loads in flight at once, and every lane sums down its own column.
// Arithmetic is the tap-list operator's: acc = acc + sample * weight in tap order, no FMA contraction - the
// results are bit-identical to it (tests/test_gpu_round5.py), so the golden fixtures g21 / g22 hold unchanged.
#pragma once
#include <hip/hip_runtime.h>

#include "wt_internal.h"
#include "wt_device.h"
#include "wt_stencil.h"

#define WT_AXIS_MAX_TAPS 33
#define WT_AXIS_SEG 1024          // output pixels of one LDS segment (4 per lane)
#define WT_AXIS_MAX_SPAN 2048     // largest (max offset - min offset) the LDS form takes

// (wt_pad_index: wt_kernels_common.h, included before this header; wt_vpack / wt_vunpack: wt_stencil.h)

template <typename T>
struct AxisArgs {
    const T *in;
    T *out;
    int W, P, Y, Z;      // a (Z, Y, X = W) cube stored as (Z * Y) rows of pitch P; images: Z = 1
    int n;               // taps
    int omin, span;      // x form: smallest offset, largest - smallest
    int o0, step;        // march: offset of tap 0, distance between consecutive taps (> 0)
    int mode, dil;       // WT_PAD_*, dilation of the polyphase modes
    T cval;              // WT_PAD_CONSTANT
    int axis;            // march: 1 = y (inside every slice), 0 = z (across slices)
    int S, chunks;       // march: chain steps per work item, work items per chain
    int o[WT_AXIS_MAX_TAPS];
    T w[WT_AXIS_MAX_TAPS];
};

// along x: out[row][x] = sum_j w_j * in[row][pad(x + o_j)]
template <typename T>
__global__ __launch_bounds__(256) void wt_axis_x_kernel(AxisArgs<T> a)
{
#pragma clang fp contract(off)
    __shared__ T seg[WT_AXIS_SEG + WT_AXIS_MAX_SPAN];
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * WT_AXIS_SEG;
    const int nload = min(WT_AXIS_SEG, a.W - x0) + a.span;
    for (int row = blockIdx.y; row < a.Z * a.Y; row += gridDim.y) {
        const T *src = a.in + (int64_t)row * a.P;
        for (int i = tid; i < nload; i += 256) {
            const int xi = wt_pad_index(x0 + a.omin + i, a.W, a.mode, a.dil);
            seg[i] = xi < 0 ? a.cval : src[xi];
        }
        __syncthreads();
        T *dst = a.out + (int64_t)row * a.P;
#pragma unroll
        for (int k = 0; k < WT_AXIS_SEG / 256; ++k) {
            const int xl = tid + 256 * k;
            if (x0 + xl < a.W) {
                T acc = (T)0;
                for (int j = 0; j < a.n; ++j) acc = acc + seg[xl + a.o[j] - a.omin] * a.w[j];
                dst[x0 + xl] = acc;
            }
        }
        __syncthreads();
    }
}

// along y or z: a workgroup owns 64 columns and one chunk of S steps of one polyphase chain of one line set.
// Chain element e of the INPUT sits at position q + o0 + step * e of the axis (border rule applied); the output
// of step r, at position q + step * r, needs elements r .. r + K - 1.  The S + K - 1 input rows of the chunk are
// staged in LDS by the four waves at once (every load of the chunk is in flight together: the marching form
// with one row per step had 250 waves for a 2048^2 image and sat on its load latency), then every lane sums its
// outputs from its own LDS column - conflict-free, no second barrier.
template <typename T>
__global__ __launch_bounds__(256) void wt_axis_tile_kernel(AxisArgs<T> a)
{
#pragma clang fp contract(off)
    extern __shared__ __align__(16) unsigned char wt_axis_tile_raw[];
    T *tile = reinterpret_cast<T *>(wt_axis_tile_raw);          // [S + K - 1][64]
    const int px = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + px;
    const int K = a.n, d = a.step;
    const int n_axis = a.axis == 1 ? a.Y : a.Z;                 // samples along the filtered axis
    const int n_outer = a.axis == 1 ? a.Z : a.Y;                // independent lines per column
    const int phases = min(d, n_axis);
    // work item -> (outer line, chain phase q, chunk c)
    int item = blockIdx.y;
    const int c = item % a.chunks;
    item /= a.chunks;
    const int q = item % phases;
    const int outer = item / phases;
    if (outer >= n_outer) return;                               // (whole workgroup)
    const int n_q = (n_axis - q + d - 1) / d;                   // chain length
    const int r0 = c * a.S, r1 = min(r0 + a.S, n_q);
    if (r0 >= r1) return;
    const bool lane_ok = x < a.W;
    auto row_of = [&](int p) -> int64_t { return a.axis == 1 ? (int64_t)outer * a.Y + p : (int64_t)p * a.Y + outer; };
    const int n_in = (r1 - r0) + K - 1;
    for (int e = g; e < n_in; e += 4) {
        const int p = wt_pad_index(q + a.o0 + d * (r0 + e), n_axis, a.mode, a.dil);
        T v = a.cval;
        if (p >= 0 && lane_ok) v = a.in[row_of(p) * a.P + x];
        tile[e * 64 + px] = v;
    }
    __syncthreads();
    if (!lane_ok) return;
    for (int r = r0 + g; r < r1; r += 4) {
        T acc = (T)0;
        const T *col = tile + (r - r0) * 64 + px;
        for (int j = 0; j < K; ++j) acc = acc + col[j * 64] * a.w[j];
        a.out[row_of(q + d * r) * a.P + x] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// host side (both element types)
// ---------------------------------------------------------------------------------------------
extern int g_opt_axis_filter;      // wt_set_option("axis_filter", 0/1); defined in wt_apps.hip

// 0: launched; -1: this tap set / axis is not one the tiled kernels take (the caller falls back to the tap-list
// operator); > 0: error.  axis: 2 = x, 1 = y, 0 = z (cubes).
template <typename T>
static int wt_axis_filter_launch(wt_ctx *c, const T *in, T *out, int W, int P, int nrows, int depth, int axis, const int32_t *offs,
                                 const T *wts, int n, int mode, T cval, int dil)
{
    constexpr int PX = WtVec<T>::PX;
    if (!g_opt_axis_filter || n < 1 || n > WT_AXIS_MAX_TAPS) return -1;
    const int Z = depth > 0 ? depth : 1, Y = nrows / Z;
    if (axis == 0 && depth <= 0) return -1;
    AxisArgs<T> a{};
    a.in = in; a.out = out; a.W = W; a.P = P; a.Y = Y; a.Z = Z; a.n = n; a.mode = mode; a.dil = dil; a.cval = cval;
    int omin = offs[0], omax = offs[0];
    for (int j = 0; j < n; ++j) {
        a.o[j] = offs[j];
        a.w[j] = wts[j];
        omin = std::min(omin, (int)offs[j]);
        omax = std::max(omax, (int)offs[j]);
    }
    if (axis == 2) {
        if ((int64_t)omax - omin > WT_AXIS_MAX_SPAN) return -1;
        a.omin = omin;
        a.span = omax - omin;
        ProfScope ps(c, sizeof(T) == 8 ? "wt64_axis_x_kernel" : "wt_axis_x_kernel");
        hipLaunchKernelGGL(wt_axis_x_kernel<T>, dim3((W + WT_AXIS_SEG - 1) / WT_AXIS_SEG, (unsigned)std::min(nrows, 32768)), dim3(256), 0, c->stream, a);
        WT_HIP(hipGetLastError());
        return 0;
    }
    // the march needs the offsets in an arithmetic progression with a positive step
    const int step = n > 1 ? offs[1] - offs[0] : 1;
    if (step < 1) return -1;
    for (int j = 1; j < n; ++j)
        if (offs[j] - offs[j - 1] != step) return -1;
    const int n_axis = axis == 1 ? Y : Z, n_outer = axis == 1 ? Z : Y;
    const int phases = std::min(step, n_axis);
    const int n_max = (n_axis + step - 1) / step;                // longest chain
    const int xblocks = (W + 63) / 64;
    // chunks of up to 64 chain steps (the K - 1 halo rows of a chunk are loaded again by its neighbour): at least
    // 4 K steps where the chains are long enough, fewer - down to the whole chain - where they are short
    (void)PX;
    const int s_cap = (int)(((size_t)60 << 10) / (64 * sizeof(T))) - (n - 1);        // rows of a chunk that fit 60 KiB of LDS
    int S = std::min({n_max, std::max(64, 4 * n), s_cap});
    int chunks = (n_max + S - 1) / S;
    while ((int64_t)n_outer * phases * chunks > 65535) {
        if (chunks == 1) return -1;                              // (more lines than the grid holds: tap-list operator)
        S *= 2;
        chunks = (n_max + S - 1) / S;
    }
    const size_t lds = (size_t)(S + n - 1) * 64 * sizeof(T);
    if (lds > (size_t)60 << 10) return -1;
    a.o0 = offs[0]; a.step = step; a.axis = axis; a.S = S; a.chunks = chunks;
    ProfScope ps(c, sizeof(T) == 8 ? "wt64_axis_tile_kernel" : "wt_axis_tile_kernel");
    hipLaunchKernelGGL(wt_axis_tile_kernel<T>, dim3(xblocks, (unsigned)(n_outer * phases * chunks)), dim3(256), lds, c->stream, a);
    WT_HIP(hipGetLastError());
    return 0;
}
