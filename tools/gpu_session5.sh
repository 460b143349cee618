#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s5; rm -rf $O; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -12 $O/pytest_gpu.log
timeout -k 10 300 python bench.py --config cfg5 --no-build --no-cpu > $O/bench_cfg5.json 2> $O/bench_cfg5.err; python - <<'P'
import json
d=json.load(open('gpurun_out/s5/bench_cfg5.json')); print(d['value'], d['ms_per_step']); [print('  ',k,v) for k,v in d['kernels'].items()]
P
