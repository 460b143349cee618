#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s10; rm -rf $O; mkdir -p $O
REPS=3 timeout -k 10 800 tools/try_variants.sh > $O/variants.txt 2>&1
cat $O/variants.txt
