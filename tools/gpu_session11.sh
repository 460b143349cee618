#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s11; rm -rf $O; mkdir -p $O
for sp in 0.25 2 0.25 5 0.25 10 0.25 2 5; do
  echo -n "spinup $sp: "; python bench.py --brief --steps 30 --no-build --spinup $sp
done 2>&1 | tee $O/spinup.txt
echo "--- steps 200"
python bench.py --brief --steps 200 --no-build --spinup 0.25 | tee -a $O/spinup.txt
python bench.py --brief --steps 2000 --no-build --spinup 0.25 | tee -a $O/spinup.txt
