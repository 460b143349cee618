#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s12; rm -rf $O; mkdir -p $O
for i in 1 2 3 4 5 6; do
  echo -n "separate: "; python bench.py --brief --steps 30 --no-build
  echo -n "arena12: "; WT_ARENA=12 python bench.py --brief --steps 30 --no-build
  echo -n "arena12 skew0: "; WT_ARENA=12 WT_PLANE_SKEW=0 python bench.py --brief --steps 30 --no-build
  echo -n "arena12 skew 69632: "; WT_ARENA=12 WT_PLANE_SKEW=69632 python bench.py --brief --steps 30 --no-build
done 2>&1 | tee $O/arena.txt
