#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s17; rm -rf $O; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log
for c in cfg3 cfg3 cfg5; do timeout -k 10 400 python bench.py --config $c --no-build --no-cpu --brief; done 2>&1 | tee $O/configs.txt
