#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s14; rm -rf $O; mkdir -p $O
WT_SCATTER=12 timeout -k 10 120 python -c "import __graft_entry__ as e; e.smoke()" 2>&1 | tail -2
for i in 1 2 3 4; do
  echo -n "separate: "; python bench.py --brief --steps 30 --no-build
  echo -n "arena12: "; WT_ARENA=12 python bench.py --brief --steps 30 --no-build
  echo -n "scatter1: "; WT_SCATTER=1 python bench.py --brief --steps 30 --no-build
  echo -n "scatter12: "; WT_SCATTER=12 python bench.py --brief --steps 30 --no-build
  echo -n "scatter4: "; WT_SCATTER=4 python bench.py --brief --steps 30 --no-build
done 2>&1 | tee $O/scatter.txt
