#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s23; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_round2.py tests/test_gpu_strips.py tests/test_gpu_parity.py -m gpu -x -q -k "fast or bench_step or decompose_sum or split or sharded or cfg or large or strip or 8192 or transform or sweep" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for rep in 1 2; do
  echo -n "auto: "; python bench.py --brief --steps 30 --no-build --no-cpu
  echo -n "off: "; WT_FUSED_WPC1=0 python bench.py --brief --steps 30 --no-build --no-cpu
  echo -n "two-call auto: "; python bench.py --brief --steps 30 --no-build --no-cpu --two-call
  echo -n "two-call off: "; WT_FUSED_WPC1=0 python bench.py --brief --steps 30 --no-build --no-cpu --two-call
  echo -n "cfg3 auto: "; python bench.py --config cfg3 --brief --no-build --no-cpu
  echo -n "cfg3 off: "; WT_FUSED_WPC1=0 python bench.py --config cfg3 --brief --no-build --no-cpu
  echo -n "cfg2 auto: "; python bench.py --config cfg2 --brief --no-build --no-cpu
done 2>&1 | tee $O/wpc1.txt
