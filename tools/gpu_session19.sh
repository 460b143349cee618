#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s19; rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "bilateral or wow or row_kernel or lattice or chain or cfg or strip or operators" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
for i in 1 2; do
  echo -n "new: "; python bench.py --config cfg5 --no-build --no-cpu --brief
  echo -n "old(rowpd2 lib): "; WATROO_HIP_LIB=$PWD/variants/rowpd2.so python bench.py --config cfg5 --no-build --no-cpu --brief
done 2>&1 | tee $O/cfg5.txt
