// Fused multi-scale passes (several consecutive scales per HBM round trip).
#pragma once
#include "wt_internal.h"

#define WT_FUSED_MAX_SCALES 3
#define WT_FUSED_MAX_FIRST_SCALE 3

static inline bool wt_fused_supported(const wt_plan *) { return false; }

static inline int wt_fused_launch(wt_plan *, const float *, float *, float **, int, int)
{
    WT_FAIL("fused passes not built");
}
