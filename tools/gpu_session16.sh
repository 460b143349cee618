#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s16; rm -rf $O; mkdir -p $O
for i in 1 2; do
for kb in 2048 512 128 8192 32768 65536; do
  echo -n "chunk ${kb} KB: "; WT_SCATTER_CHUNK_KB=$kb python bench.py --brief --steps 30 --no-build
done
for sc in 1 2 8 12; do
  echo -n "scatter group $sc: "; WT_SCATTER=$sc python bench.py --brief --steps 30 --no-build
done
done 2>&1 | tee $O/chunks.txt
