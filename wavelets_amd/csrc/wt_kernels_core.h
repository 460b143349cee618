// Kernels of the core unit (wt_core.hip): device-to-device copies of planes mapped over scattered chunks, fill.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "wt_internal.h"
#include "wt_device.h"
#include "wt_stencil.h"

// dst[r][0..cols) = src[r][0..cols) for r < rows (pitches in floats): the device-to-device copies of
// planes whose memory is a mapping of scattered physical chunks (hipMemcpy2D refuses those)
__global__ __launch_bounds__(256) void wt_copy2d_kernel(float *dst, int64_t dpitch, const float *src, int64_t spitch,
                                                       int cols, int rows)
{
    for (int r = blockIdx.y; r < rows; r += gridDim.y)
        for (int x = blockIdx.x * 256 + threadIdx.x; x < cols; x += gridDim.x * 256)
            dst[(int64_t)r * dpitch + x] = src[(int64_t)r * spitch + x];
}

__global__ __launch_bounds__(256) void wt_copy_kernel(float *dst, const float *src, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x)
        reinterpret_cast<float4 *>(dst)[i] = reinterpret_cast<const float4 *>(src)[i];
}

__global__ __launch_bounds__(256) void wt_fill_kernel(float *dst, int64_t n4, float value)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4;
         i += (int64_t)gridDim.x * blockDim.x)
        reinterpret_cast<float4 *>(dst)[i] = make_float4(value, value, value, value);
}

