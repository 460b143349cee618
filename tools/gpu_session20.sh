#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s20; rm -rf $O; mkdir -p $O
timeout -k 10 900 python tools/fuzz.py 90 7 > $O/fuzz.txt 2>&1; tail -8 $O/fuzz.txt
timeout -k 10 300 python -m pytest tests/test_gpu_strips.py -m gpu -x -q > $O/pytest_strips.log 2>&1; tail -3 $O/pytest_strips.log
