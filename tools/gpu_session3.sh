#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s3; rm -rf $O; mkdir -p $O
timeout -k 10 120 tools/membench2 > $O/membench2.txt 2>&1
grep "PERM\|D  1  WGs  512\|D  8  WGs  512" $O/membench2.txt
cat > $O/envs.txt <<'E2'
abl4 WATROO_HIP_LIB=variants/abl.so WT_FUSED_DEBUG=4
abl1 WATROO_HIP_LIB=variants/abl.so WT_FUSED_DEBUG=1
abl2 WATROO_HIP_LIB=variants/abl.so WT_FUSED_DEBUG=2
abl3 WATROO_HIP_LIB=variants/abl.so WT_FUSED_DEBUG=3
abl6 WATROO_HIP_LIB=variants/abl.so WT_FUSED_DEBUG=6
rounds2 WT_FUSED_ROUNDS=2
E2
VARIANT_ENVS=$O/envs.txt REPS=2 timeout -k 10 600 tools/try_variants.sh > $O/variants.txt 2>&1
cat $O/variants.txt
