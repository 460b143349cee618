// Exact median of |x| on double planes: the kernels of wt64_abs_median (histogram levels, the windowed first level,
// the gathered candidate list and its single-workgroup finish).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include "wt_internal.h"
#include "wt_device.h"
#include "wt_math64.h"
#include "wt_kernels_common.h"

// Exact median of |x| (np.median(np.abs(data[0])), watroo/wavelets.py:127): non-negative doubles
// order like their bit patterns; radix select over 63 bits, 11 bits per pass (LDS-privatised bins),
// selection state on the device like the float32 select.
struct Select64State {
    unsigned long long k, cum_le, prefix;
    uint32_t failed, pad;         // failed: 1 = rank not found (NaN input), 2 = candidate list too small (fall back to the radix passes)
    unsigned long long bin_count; // elements in the bin the last step selected (what a collect pass would gather)
    unsigned long long upper;     // list select: smallest key above the lower median (~0: not among the candidates)
};

// Read stream over the valid samples of a plane for the select kernels: work items are (row, chunk of
// 256 * 4 double2) pairs dealt round-robin to the blocks; the four 16-byte loads of the NEXT item are
// issued before the current item is consumed (8 loads in flight per thread - the one-item loop of round
// 3 streamed at 4.2 TB/s, with a full drain between a row's chunks).  f(key[8], ok[8]) sees the 63-bit
// magnitude keys of a thread's eight samples of an item and which of them are valid samples.
template <typename F>
__device__ __forceinline__ void wt64_scan_keys(const double *p, int nrows, int P, int W, F &&f)
{
    constexpr int U = 4;
    const int X2 = (W + 1) / 2;                          // double2 groups per row (rows are 16-byte aligned: P even)
    const int nchunk = (X2 + 256 * U - 1) / (256 * U);
    const int64_t nitems = (int64_t)nrows * nchunk;
    auto load = [&](int64_t item, double2 (&v)[U]) {
        const int r = (int)(item / nchunk), c = (int)(item - (int64_t)r * nchunk);
        const double *row = p + (int64_t)r * P;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            typedef double nt2d __attribute__((ext_vector_type(2)));
            const nt2d t = __builtin_nontemporal_load(reinterpret_cast<const nt2d *>(row + 2 * min(c * 256 * U + 256 * u + (int)threadIdx.x, X2 - 1)));
            v[u] = make_double2(t.x, t.y);
        }
    };
    auto consume = [&](int64_t item, const double2 (&v)[U]) {
        const int c = (int)(item % nchunk);
        unsigned long long key[2 * U];
        bool ok[2 * U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int xx = c * 256 * U + 256 * u + (int)threadIdx.x;
            const double e[2] = {v[u].x, v[u].y};
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                key[2 * u + k] = (unsigned long long)__double_as_longlong(e[k]) & 0x7fffffffffffffffull;
                ok[2 * u + k] = xx < X2 && 2 * xx + k < W;
            }
        }
        f(key, ok);
    };
    double2 va[U], vb[U];
    int64_t item = blockIdx.x;
    if (item < nitems) load(item, va);
    while (item < nitems) {                              // two items per trip: no register copies
        const int64_t i1 = item + gridDim.x, i2 = i1 + gridDim.x;
        if (i1 < nitems) load(i1, vb);
        consume(item, va);
        if (i1 >= nitems) break;
        if (i2 < nitems) load(i2, va);
        consume(i1, vb);
        item = i2;
    }
}

__global__ __launch_bounds__(256) void wt64_hist_kernel(const double *p, int nrows, int P, int W, unsigned long long prefix_mask,
                                                        const Select64State *st, int shift, uint32_t bin_mask, uint32_t *hist)
{
    const unsigned long long prefix_val = st->prefix & prefix_mask;
    __shared__ uint32_t lh[WT_HIST_BINS];
    for (int i = threadIdx.x; i < WT_HIST_BINS; i += 256) lh[i] = 0;
    __syncthreads();
    wt64_scan_keys(p, nrows, P, W, [&](const unsigned long long (&w)[8], const bool (&ok)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (ok[j] && (w[j] & prefix_mask) == prefix_val) atomicAdd(&lh[(uint32_t)(w[j] >> shift) & bin_mask], 1u);
    });
    __syncthreads();
    for (int i = threadIdx.x; i < WT_HIST_BINS; i += 256)
        if (lh[i]) atomicAdd(&hist[i], lh[i]);
}

__global__ __launch_bounds__(256) void wt64_select_step_kernel(uint32_t *hist, Select64State *st, int nbins, int shift, int last)
{
    __shared__ unsigned long long part[256];
    const int per = (nbins + 255) / 256;
    const int b0 = threadIdx.x * per;
    uint32_t h[WT_HIST_BINS / 256];
    unsigned long long s = 0;
#pragma unroll
    for (int i = 0; i < WT_HIST_BINS / 256; ++i) {
        h[i] = (i < per && b0 + i < nbins) ? hist[b0 + i] : 0u;
        s += h[i];
    }
    const unsigned long long k = st->k, cum_le = st->cum_le, prefix = st->prefix;
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const unsigned long long v = threadIdx.x >= off ? part[threadIdx.x - off] : 0ull;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    const unsigned long long incl = part[threadIdx.x];
    unsigned long long cum = incl - s;
    if (k >= cum && k < incl) {
#pragma unroll
        for (int i = 0; i < WT_HIST_BINS / 256; ++i) {
            if (k < cum + h[i]) {
                st->k = k - cum;
                st->cum_le = cum_le + cum + (last ? h[i] : 0);
                st->prefix = prefix | ((unsigned long long)(b0 + i) << shift);
                st->bin_count = h[i];
                break;
            }
            cum += h[i];
        }
    }
    if (threadIdx.x == 255 && k >= incl && st->failed == 0) st->failed = 1;
    for (int i = threadIdx.x; i < nbins; i += 256) hist[i] = 0;
}

// Round 4: after two radix levels (22 of the 63 bits) the selected bin of a continuous plane holds a
// few ten thousand elements (8192^2 Gaussian: ~2e-4 of them).  Instead of four more passes over the
// plane, ONE pass gathers those elements' keys into a list (this kernel) and one workgroup finishes
// the select on the list (below) - lower AND upper median, so the extra pass for the upper median of an
// even count disappears too.  A bin that does not fit the list (ties: constant or quantised data) is
// left alone: bin_count > cap, nothing is gathered, the host continues with the radix passes.
// step after a WINDOWED riding histogram (22-bit keys base + bin in bins 1 .. 2046): as
// wt_select_window_step_kernel; failed = 3 when the rank lies outside the window
__global__ __launch_bounds__(256) void wt64_select_window_step_kernel(uint32_t *hist, Select64State *st, const uint32_t *base)
{
    __shared__ unsigned long long part[256];
    constexpr int per = WT_HIST_BINS / 256;
    const int b0 = threadIdx.x * per;
    uint32_t h[per];
    unsigned long long s = 0;
#pragma unroll
    for (int i = 0; i < per; ++i) {
        h[i] = hist[b0 + i];
        s += h[i];
    }
    const unsigned long long k = st->k;
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {
        const unsigned long long v = threadIdx.x >= off ? part[threadIdx.x - off] : 0ull;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    const unsigned long long incl = part[threadIdx.x];
    unsigned long long cum = incl - s;
    if (k >= cum && k < incl) {
#pragma unroll
        for (int i = 0; i < per; ++i) {
            if (k < cum + h[i]) {
                const int bin = b0 + i;
                if (bin == 0 || bin == WT_HIST_BINS - 1) {
                    st->failed = 3;
                } else {
                    st->k = k - cum;
                    st->cum_le = cum;
                    st->prefix = (unsigned long long)(*base + (uint32_t)bin) << 41;
                    st->bin_count = h[i];
                }
                break;
            }
            cum += h[i];
        }
    }
    if (threadIdx.x == 255 && k >= incl && st->failed == 0) st->failed = 1;
    for (int i = threadIdx.x; i < WT_HIST_BINS; i += 256) hist[i] = 0;
}

__global__ __launch_bounds__(256) void wt64_collect_kernel(const double *p, int nrows, int P, int W, unsigned long long prefix_mask,
                                                           const Select64State *st, unsigned long long *list, unsigned long long cap)
{
    if (st->bin_count > cap || st->failed) return;
    const unsigned long long prefix_val = st->prefix & prefix_mask;
    unsigned long long *count = list + cap;              // the counter word sits behind the list
    // Hits are staged in LDS and leave with ONE global atomic per block: appends through a single global
    // counter serialise (~6 ns each - 14 000 of them doubled the time of this pass at 8192^2).  A block
    // that overflows its stage (bins near the list's capacity) appends the rest directly.
    constexpr int STAGE = 1024;
    __shared__ unsigned long long stage[STAGE];
    __shared__ unsigned int n_staged;
    __shared__ unsigned long long base;
    if (threadIdx.x == 0) n_staged = 0;
    __syncthreads();
    wt64_scan_keys(p, nrows, P, W, [&](const unsigned long long (&w)[8], const bool (&ok)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (ok[j] && (w[j] & prefix_mask) == prefix_val) {
                const unsigned int i = atomicAdd(&n_staged, 1u);
                if (i < STAGE) {
                    stage[i] = w[j];
                } else {
                    const unsigned long long slot = atomicAdd(count, 1ull);
                    if (slot < cap) list[slot] = w[j];
                }
            }
    });
    __syncthreads();
    const unsigned int n = min(n_staged, (unsigned int)STAGE);
    if (threadIdx.x == 0 && n) base = atomicAdd(count, (unsigned long long)n);
    __syncthreads();
    for (unsigned int i = threadIdx.x; i < n; i += 256)
        if (base + i < cap) list[base + i] = stage[i];
}

// One workgroup finishes the select on the gathered keys: radix levels over the remaining `bits_left`
// bits (11 per level, LDS bins), then the smallest key above the result (the upper median of an even
// count).  State in / out as wt64_select_step_kernel; failed = 2 when nothing was gathered.
__global__ __launch_bounds__(1024) void wt64_list_select_kernel(const unsigned long long *list, unsigned long long cap, Select64State *st,
                                                                int bits_left)
{
    __shared__ uint32_t lh[WT_HIST_BINS];
    __shared__ unsigned long long part[1024];
    __shared__ unsigned long long sh_k, sh_cum, sh_prefix;
    __shared__ int sh_found;
    if (st->failed) return;
    const unsigned long long n = list[cap];
    if (st->bin_count > cap || n != st->bin_count) {     // (not gathered, or the counts disagree: let the host fall back)
        if (threadIdx.x == 0) st->failed = 2;
        return;
    }
    unsigned long long k = st->k, cum_le = st->cum_le, prefix = st->prefix;
    unsigned long long known = ~((1ull << bits_left) - 1ull) & 0x7fffffffffffffffull;   // bits fixed so far
    int left = bits_left;
    while (left > 0) {
        const int nb = left >= 11 ? 11 : left, shift = left - nb, nbins = 1 << nb;
        for (int i = threadIdx.x; i < WT_HIST_BINS; i += 1024) lh[i] = 0;
        __syncthreads();
        for (unsigned long long i = threadIdx.x; i < n; i += 1024) {
            const unsigned long long w = list[i];
            if ((w & known) == (prefix & known)) atomicAdd(&lh[(uint32_t)(w >> shift) & (uint32_t)(nbins - 1)], 1u);
        }
        __syncthreads();
        // two bins per thread, inclusive scan over the threads' sums
        const uint32_t h0 = lh[2 * threadIdx.x], h1 = lh[2 * threadIdx.x + 1];
        part[threadIdx.x] = (unsigned long long)h0 + h1;
        if (threadIdx.x == 0) sh_found = 0;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {
            const unsigned long long v = threadIdx.x >= off ? part[threadIdx.x - off] : 0ull;
            __syncthreads();
            part[threadIdx.x] += v;
            __syncthreads();
        }
        const unsigned long long incl = part[threadIdx.x], excl = incl - h0 - h1;
        if (k >= excl && k < incl) {
            const bool second = k >= excl + h0;
            const unsigned long long cum = excl + (second ? h0 : 0);
            sh_k = k - cum;
            sh_cum = cum_le + cum + (shift == 0 ? (second ? h1 : h0) : 0);
            sh_prefix = prefix | ((unsigned long long)(2 * threadIdx.x + (second ? 1 : 0)) << shift);
            sh_found = 1;
        }
        __syncthreads();
        if (!sh_found) {
            if (threadIdx.x == 0) st->failed = 1;
            return;
        }
        k = sh_k; cum_le = sh_cum; prefix = sh_prefix;
        known |= (unsigned long long)(nbins - 1) << shift;
        left -= nb;
        __syncthreads();
    }
    // smallest gathered key above the result
    unsigned long long best = ~0ull;
    for (unsigned long long i = threadIdx.x; i < n; i += 1024) {
        const unsigned long long w = list[i];
        if (w > prefix && w < best) best = w;
    }
    part[threadIdx.x] = best;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (threadIdx.x < off && part[threadIdx.x + off] < part[threadIdx.x]) part[threadIdx.x] = part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        st->k = k; st->cum_le = cum_le; st->prefix = prefix;
        st->upper = part[0];
    }
}

__global__ __launch_bounds__(256) void wt64_min_greater_kernel(const double *p, int nrows, int P, int W, unsigned long long than,
                                                               unsigned long long *result)
{
    unsigned long long best = ~0ull;
    wt64_scan_keys(p, nrows, P, W, [&](const unsigned long long (&w)[8], const bool (&ok)[8]) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (ok[j] && w[j] > than && w[j] < best) best = w[j];
    });
    // one global atomic per block
    __shared__ unsigned long long wb[256];
    wb[threadIdx.x] = best;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off && wb[threadIdx.x + off] < wb[threadIdx.x]) wb[threadIdx.x] = wb[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0 && wb[0] != ~0ull) atomicMin(result, wb[0]);
}

