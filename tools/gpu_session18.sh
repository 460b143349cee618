#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/s18; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "median or cfg3 or rccl or noise or coefficients_methods" > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
for c in cfg3 cfg3; do timeout -k 10 400 python bench.py --config $c --no-build --no-cpu --brief; done 2>&1 | tee $O/configs.txt
python - <<'P'
from wavelets_amd import _lib as L
import numpy as np
ctx=L.default_context(); p=L.Plan(ctx,8192,8192,L.B3SPLINE,0); p.upload(0, np.random.default_rng(0).standard_normal((8192,8192),dtype=np.float32))
p.abs_median(0); ctx.profile(True); ctx.profile_reset()
import time; t=time.perf_counter(); 
for _ in range(5): p.abs_median(0)
print("abs_median wall ms", (time.perf_counter()-t)/5*1e3, ctx.profile_entries())
P
