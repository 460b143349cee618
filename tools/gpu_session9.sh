#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s9; rm -rf $O; mkdir -p $O
REPS=3 timeout -k 10 600 tools/try_variants.sh > $O/variants.txt 2>&1
cat $O/variants.txt
WATROO_HIP_LIB=$PWD/variants/peel.so timeout -k 10 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_strips.py -m gpu -x -q -k "fast or bench_step or decompose_sum or split or sharded" > $O/pytest_peel.log 2>&1; tail -3 $O/pytest_peel.log
