#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s13; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for pad in 0 256 4096 16384 65536 262144 1048576 1052672 2097152 2101248 4198400 8392704 16781312 33558528 67112960 134221824 17825792 35651584; do
  echo -n "arena pad $pad skew0: "; WT_ARENA=12 WT_PLANE_SKEW=0 WT_ARENA_PAD=$pad python bench.py --brief --steps 30 --no-build
done
done 2>&1 | tee $O/arena_pad.txt
