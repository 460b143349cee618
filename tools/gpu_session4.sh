#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s4; rm -rf $O; mkdir -p $O
timeout -k 10 120 tools/membench3 > $O/membench3.txt 2>&1; cat $O/membench3.txt
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py -m gpu -x -q -k "denoise_between or with_sum or richardson or negative" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for c in cfg3; do timeout -k 10 400 python bench.py --config $c --no-build --no-cpu > $O/bench_$c.json 2> $O/bench_$c.err; cut -c1-200 $O/bench_$c.json; tail -3 $O/bench_$c.err; done
