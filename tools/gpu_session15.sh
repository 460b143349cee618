#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s15; rm -rf $O; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -8 $O/pytest_gpu.log
python - <<'P'
import time, numpy as np
import __graft_entry__ as e
from wavelets_amd import _lib as L
ctx = L.default_context()
for H in (8192, 4096, 512):
    t=time.perf_counter(); p = L.Plan(ctx, H, H, L.B3SPLINE, 6); ctx.sync(); t1=time.perf_counter()-t
    t=time.perf_counter(); p.close(); t2=time.perf_counter()-t
    print(f"plan create {H}^2 L=6: {t1*1e3:.1f} ms, destroy {t2*1e3:.1f} ms")
P
for i in 1 2 3; do
  echo -n "default(scatter4): "; python bench.py --brief --steps 30 --no-build
  echo -n "WT_SCATTER=0: "; WT_SCATTER=0 python bench.py --brief --steps 30 --no-build
done 2>&1 | tee $O/scatter.txt
for c in cfg2 cfg3 cfg5; do timeout -k 10 400 python bench.py --config $c --no-build --no-cpu --brief; WT_SCATTER=0 timeout -k 10 400 python bench.py --config $c --no-build --no-cpu --brief; done 2>&1 | tee $O/configs.txt
