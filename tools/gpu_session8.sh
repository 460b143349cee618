#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s8; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_strips.py -m gpu -x -q -k "fast or bench_step or decompose_sum or split or sharded" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
REPS=3 timeout -k 10 600 tools/try_variants.sh > $O/variants.txt 2>&1
cat $O/variants.txt
