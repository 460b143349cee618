// One instantiation group of the fused passes per translation unit (see wt_fused_decl.h).  Compiled
// by __graft_entry__.build() once per (WT_TU_F64, WT_TU_K, WT_TU_ACC):
//     hipcc -c wt_fused_tu.hip -DWT_TU_F64=0 -DWT_TU_K=5 -DWT_TU_ACC=1 -o _build/fused_f32_k5_acc1.o
#include "wt_fused.h"
#include "wt_unit_probe.h"

WT_UNIT_PROBE_DEFINE

#if !defined(WT_TU_F64) || !defined(WT_TU_K) || !defined(WT_TU_ACC)
#error "wt_fused_tu.hip: define WT_TU_F64 (0/1), WT_TU_K (3/5) and WT_TU_ACC (0..3)"
#endif

#define WT_TU_CAT2(a, b, c, d) a##b##c##d
#define WT_TU_CAT(a, b, c, d) WT_TU_CAT2(a, b, c, d)

#if WT_TU_F64
int WT_TU_CAT(wt_fused_tu_f64_k, WT_TU_K, _acc, WT_TU_ACC)(wt_plan64 *p, const FusedArgsT<double> &a, int s0, int ns, const FusedRows &rows)
{
    return wt_fused64_dispatch_acc<WT_TU_K, WT_TU_ACC>(p, a, s0, ns, rows);
}
#else
int WT_TU_CAT(wt_fused_tu_f32_k, WT_TU_K, _acc, WT_TU_ACC)(wt_plan *p, const FusedArgs &a, int s0, int ns, const FusedRows &rows)
{
    return wt_fused_dispatch_acc<WT_TU_K, WT_TU_ACC>(p, a, s0, ns, rows);
}
#endif
