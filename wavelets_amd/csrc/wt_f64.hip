// libwatroo_hip.so - host side of the C ABI, unit 4 of 4: the float64 engine (wt_f64.h: the wt64_* entry points,
// their generic kernels, the launches of the fused double passes and of the float64 stencil / bilateral unit).
// gfx950 only.  Compiled with -DWT_TU_NAME=f64 (wt_math64.h).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "wt_host.h"
#include "wt_kernels_common.h"
#include "wt_fused_decl.h"
#include "wt_fft.h"
#include "wt_axis.h"
#include "wt_f64.h"
#include "wt_unit_probe.h"

WT_UNIT_PROBE_DEFINE
