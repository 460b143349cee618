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
#define WT_AXIS_R 8               // outputs a lane of the tile kernel sums side by side

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
    constexpr int NV = (WT_AXIS_SEG + WT_AXIS_MAX_SPAN) / 256;   // segment elements a lane stages at most
    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * WT_AXIS_SEG;
    const int nload = min(WT_AXIS_SEG, a.W - x0) + a.span;
    const int nrows = a.Z * a.Y;
    // the columns this lane stages are the same on every row: border rule applied once (-1: the fill value,
    // -2: beyond the segment)
    int xi[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int i = tid + 256 * k;
        xi[k] = i < nload ? wt_pad_index(x0 + a.omin + i, a.W, a.mode, a.dil) : -2;
    }
    // a row's loads are issued together, one row ahead of the sums (the workgroup walks rows blockIdx.y,
    // + gridDim.y, ...): their latency runs under the previous row's arithmetic
    T nx[NV];
    auto fetch = [&](int row) {
        const T *src = a.in + (int64_t)row * a.P;
#pragma unroll
        for (int k = 0; k < NV; ++k) nx[k] = xi[k] >= 0 ? src[xi[k]] : a.cval;
    };
    if ((int)blockIdx.y < nrows) fetch(blockIdx.y);
    for (int row = blockIdx.y; row < nrows; row += gridDim.y) {
#pragma unroll
        for (int k = 0; k < NV; ++k)
            if (xi[k] != -2) seg[tid + 256 * k] = nx[k];
        __syncthreads();
        if (row + (int)gridDim.y < nrows) fetch(row + gridDim.y);
        T *dst = a.out + (int64_t)row * a.P;
        // the four outputs of a lane side by side, tap by tap: four independent accumulation chains (each in tap
        // order, as the tap-list operator sums) instead of one LDS round trip per tap per output
        T acc[WT_AXIS_SEG / 256];
#pragma unroll
        for (int k = 0; k < WT_AXIS_SEG / 256; ++k) acc[k] = (T)0;
        for (int j = 0; j < a.n; ++j) {
            const T *sj = seg + tid + (a.o[j] - a.omin);
            const T w = a.w[j];
#pragma unroll
            for (int k = 0; k < WT_AXIS_SEG / 256; ++k) acc[k] = acc[k] + sj[256 * k] * w;
        }
#pragma unroll
        for (int k = 0; k < WT_AXIS_SEG / 256; ++k)
            if (x0 + tid + 256 * k < a.W) dst[x0 + tid + 256 * k] = acc[k];
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
    T *tile = reinterpret_cast<T *>(wt_axis_tile_raw);          // [S + K - 1 (+ WT_AXIS_R unwritten)][64]
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
    constexpr int U = 10;                                       // loads a lane has in flight together
    for (int e0 = g; e0 < n_in; e0 += 4 * U) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + 4 * u;
            v[u] = a.cval;
            if (e < n_in && lane_ok) {
                const int p = wt_pad_index(q + a.o0 + d * (r0 + e), n_axis, a.mode, a.dil);
                if (p >= 0) v[u] = a.in[row_of(p) * a.P + x];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (e0 + 4 * u < n_in) tile[(e0 + 4 * u) * 64 + px] = v[u];
    }
    __syncthreads();
    if (!lane_ok) return;
    // A wave owns a contiguous quarter of the chunk's outputs and takes them WT_AXIS_R at a time down its LDS
    // column: tap j of output r + i reads row r + i + j, so one new row per tap serves all WT_AXIS_R chains (a
    // sliding register window, K + R - 1 LDS reads per R outputs instead of R K) and the chains are independent -
    // each still sums in tap order.  Rows past the chunk's last (up to R below it: the launcher allocates them)
    // hold whatever LDS held; they only reach accumulators that are not stored.
    const int nout = r1 - r0, B = (nout + 3) / 4;
    const int b1 = min((g + 1) * B, nout);
    for (int r = g * B; r < b1; r += WT_AXIS_R) {
        const T *col = tile + r * 64 + px;
        T v[WT_AXIS_R], acc[WT_AXIS_R];
#pragma unroll
        for (int i = 0; i < WT_AXIS_R; ++i) {
            v[i] = col[i * 64];
            acc[i] = (T)0;
        }
        for (int j = 0; j < K; ++j) {
            const T nxt = col[(j + WT_AXIS_R) * 64];
            const T w = a.w[j];
#pragma unroll
            for (int i = 0; i < WT_AXIS_R; ++i) acc[i] = acc[i] + v[i] * w;
#pragma unroll
            for (int i = 0; i + 1 < WT_AXIS_R; ++i) v[i] = v[i + 1];
            v[WT_AXIS_R - 1] = nxt;
        }
#pragma unroll
        for (int i = 0; i < WT_AXIS_R; ++i)
            if (r + i < b1) a.out[row_of(q + d * (r0 + r + i)) * a.P + x] = acc[i];
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
        // one resident round of workgroups (LDS: 12 / 24 KiB each), every one walking its rows with the next row's loads in flight
        const int nxb = (W + WT_AXIS_SEG - 1) / WT_AXIS_SEG;
        const int slots = c->num_cus * (sizeof(T) == 8 ? 6 : 8);
        const int gy = std::max(1, std::min({nrows, 32768, (slots + nxb - 1) / nxb}));
        hipLaunchKernelGGL(wt_axis_x_kernel<T>, dim3(nxb, (unsigned)gy), dim3(256), 0, c->stream, a);
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
    // chunks of 64 chain steps or more (the K - 1 halo rows of a chunk are loaded again by its neighbour): at least
    // 3 K steps where the chains are long enough, fewer - down to the whole chain - where they are short
    (void)PX;
    const int s_cap = (int)(((size_t)60 << 10) / (64 * sizeof(T))) - (n - 1) - WT_AXIS_R;   // rows of a chunk that fit 60 KiB of LDS
    int S = std::min({n_max, (std::max(64, 3 * n) + 31) / 32 * 32, s_cap});     // (a multiple of 4 waves x WT_AXIS_R outputs)
    int chunks = (n_max + S - 1) / S;
    while ((int64_t)n_outer * phases * chunks > 65535) {
        if (chunks == 1) return -1;                              // (more lines than the grid holds: tap-list operator)
        S *= 2;
        chunks = (n_max + S - 1) / S;
    }
    const size_t lds = (size_t)(S + n - 1 + WT_AXIS_R) * 64 * sizeof(T);      // (+ R rows the sliding windows may read past the end)
    if (lds > (size_t)60 << 10) return -1;
    a.o0 = offs[0]; a.step = step; a.axis = axis; a.S = S; a.chunks = chunks;
    ProfScope ps(c, sizeof(T) == 8 ? "wt64_axis_tile_kernel" : "wt_axis_tile_kernel");
    hipLaunchKernelGGL(wt_axis_tile_kernel<T>, dim3(xblocks, (unsigned)(n_outer * phases * chunks)), dim3(256), lds, c->stream, a);
    WT_HIP(hipGetLastError());
    return 0;
}
