#!/bin/bash
# round-2 GPU session 1: parity tests, variant sweep (fast addressing / vmcnt padding), SQ counters
cd "$(dirname "$0")/.."
O=gpurun_out/s1; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -3 $O/pytest_gpu.log
cat > $O/envs.txt <<'E2'
nofast WT_FUSED_NO_FAST=1
old WT_FUSED_NO_FAST=1 WATROO_HIP_LIB=variants/novmpad.so
E2
VARIANT_ENVS=$O/envs.txt REPS=3 timeout -k 10 600 tools/try_variants.sh > $O/variants.txt 2>&1
cat $O/variants.txt
timeout -k 10 200 python bench.py > $O/bench.json 2> $O/bench.err
cut -c1-400 $O/bench.json
tools/pmc_sq.sh $O/pmc_fast
WT_FUSED_NO_FAST=1 WATROO_HIP_LIB=$PWD/variants/novmpad.so tools/pmc_sq.sh $O/pmc_old
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof -- python3 $OLDPWD/bench.py --steps 20 --no-cpu --no-build > $OLDPWD/$O/rocprof_bench.log 2>&1 )
ls $O
