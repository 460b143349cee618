#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s7; rm -rf $O; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log
for rep in 1 2 3; do
  echo -n "rows 8192: "; python bench.py --brief --steps 30 --no-build
  echo -n "rows 8176: "; python bench.py --brief --steps 30 --no-build --rows 8176
  echo -n "rows 8208: "; python bench.py --brief --steps 30 --no-build --rows 8208
done 2>&1 | tee $O/rows.txt
