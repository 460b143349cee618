#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s2; rm -rf $O; mkdir -p $O
timeout -k 10 120 tools/membench2 > $O/membench2.txt 2>&1
cat $O/membench2.txt | head -70
timeout -k 10 900 python -m pytest tests/test_gpu_round2.py -m gpu -x -q > $O/pytest_r2.log 2>&1; echo "pytest rc=$?" >> $O/pytest_r2.log
tail -15 $O/pytest_r2.log
for c in cfg2 cfg3 cfg5; do timeout -k 10 400 python bench.py --config $c --no-build > $O/bench_$c.json 2> $O/bench_$c.err; cut -c1-600 $O/bench_$c.json; tail -3 $O/bench_$c.err; done
