#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s21; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for ch in 0 26 34 51 68 76 102 153 204; do
  echo -n "cfg2 chunks $ch: "; WT_FUSED_CHUNKS=$ch python bench.py --config cfg2 --brief --steps 50 --no-build --no-cpu
done
done 2>&1 | tee $O/cfg2_chunks.txt
for ch in 0 28 40 56 84 112; do
  echo -n "headline chunks $ch: "; WT_FUSED_CHUNKS=$ch python bench.py --brief --steps 30 --no-build --no-cpu
done 2>&1 | tee $O/headline_chunks.txt
