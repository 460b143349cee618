// Code objects load lazily: the HIP runtime uploads a translation unit's device code the first time one of
// its kernels is launched (~3-8 ms per unit), which used to land inside a process's first transform.  Every unit
// of the library defines an empty kernel and a host function that asks the runtime for that kernel's attributes -
// which loads the unit's code object and nothing else - and a context's warm-up thread (wt_core.hip: ctx_warm)
// walks the list below while the caller is still busy creating plans and uploading its first image.
//   a unit:          #include "wt_unit_probe.h"  +  WT_UNIT_PROBE_DEFINE   (compiled with -DWT_TU_NAME=<unit>)
//   wt_core.hip:     WT_UNITS(X) - the same names __graft_entry__._units() builds (tests/test_abi_cpu.py compares)
#pragma once
#include <hip/hip_runtime.h>

#define WT_UNITS(X)                                                                                           \
    X(core) X(transform) X(apps) X(stencil32)                                                                         \
    X(fused_f32_k5_acc0) X(fused_f32_k5_acc1) X(fused_f32_k5_acc2) X(fused_f32_k5_acc3)                       \
    X(fused_f32_k3_acc0) X(fused_f32_k3_acc1) X(fused_f32_k3_acc2) X(fused_f32_k3_acc3)                       \
    X(f64) X(stencil64)                                                                                       \
    X(fused_f64_k5_acc0) X(fused_f64_k5_acc1) X(fused_f64_k5_acc2) X(fused_f64_k5_acc3)                       \
    X(fused_f64_k3_acc0) X(fused_f64_k3_acc1) X(fused_f64_k3_acc2) X(fused_f64_k3_acc3)

#define WT_PROBE_CAT2(a, b) a##b
#define WT_PROBE_CAT(a, b) WT_PROBE_CAT2(a, b)

#ifdef WT_TU_NAME
#define WT_UNIT_PROBE_DEFINE                                                                                  \
    __global__ void WT_PROBE_CAT(wt_unit_probe_kernel_, WT_TU_NAME)() {}                                      \
    int WT_PROBE_CAT(wt_unit_load_, WT_TU_NAME)()                                                             \
    {                                                                                                         \
        hipFuncAttributes at;                                                                                 \
        return (int)hipFuncGetAttributes(&at, (const void *)WT_PROBE_CAT(wt_unit_probe_kernel_, WT_TU_NAME)); \
    }
#endif
