#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s22; rm -rf $O; mkdir -p $O
for rep in 1 2; do
for ch in 0 28 24 20 32 19 14; do
  echo -n "8192 chunksA $ch: "; WT_FUSED_CHUNKS=$ch python bench.py --brief --steps 30 --no-build --no-cpu
done
done 2>&1 | tee $O/headline_chunks.txt
for sz in 4096 6144 12288 16384; do
for ch in 0 1wg; do
  if [ $ch = 1wg ]; then n=$(python -c "import math; W=$sz; nx=math.ceil(W/960); print(256//nx)"); else n=0; fi
  echo -n "size $sz chunksA $n: "; WT_FUSED_CHUNKS=$n python bench.py --size $sz --brief --steps 30 --no-build --no-cpu
done
done 2>&1 | tee $O/sizes.txt
