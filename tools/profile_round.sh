#!/bin/bash
# End-of-session evidence: GPU tests, the default bench line (headline + cfg2/3/5), rocprofv3 kernel
# stats and the FETCH_SIZE / WRITE_SIZE passes of every configuration.  Results under
# gpurun_out/final; tools/pmc_summary.py turns the PMC passes into profiles/<tag>_pmc_summary*.csv and
# profiles/traffic.json (clear the LOCAL gpurun_out/final first: gpurun merges, old runs would mix in).
# The profiled program sits directly behind `--` and is told not to build
# (no child process under the profiler's preload).   usage: bash tools/profile_round.sh [skip-tests]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
rm -rf gpurun_out/final; mkdir -p gpurun_out/final
python -c "import __graft_entry__ as e; e.build()"
if [ "$1" != "skip-tests" ]; then
  timeout -k 10 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -5 > gpurun_out/final/pytest_gpu.log
fi
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
cd /tmp && export TMPDIR=/tmp
for c in headline cfg2 cfg3 cfg5; do
  steps=20; [ $c = cfg5 ] && steps=5
  # (--brief: no PCIe leg - the pipelined host call launches the same kernels on row blocks, which
  #  would pull the per-kernel averages down - no CPU baseline, no other configs)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_$c -- python3 $R/bench.py --config $c --steps $steps --no-cpu --no-build --brief > $R/gpurun_out/final/rocprof_$c.log 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/final/pmc_fetch_$c -- python3 $R/bench.py --config $c --steps 5 --warmup 1 --no-cpu --no-build --brief > /dev/null 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/final/pmc_write_$c -- python3 $R/bench.py --config $c --steps 5 --warmup 1 --no-cpu --no-build --brief > /dev/null 2>&1
done
# cfg5 in float64 (tools/bench_wow64.py times the float32 flow first, then the float64 one; serial order so that
# the per-kernel averages are not stretched by the side stream): kernel stats and the two PMC passes
export WT_NO_WOW_OVERLAP=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_cfg5_f64 -- python3 $R/tools/bench_wow64.py 8192 3 > $R/gpurun_out/final/rocprof_cfg5_f64.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/final/pmc_cfg5_f64/fetch -- python3 $R/tools/bench_wow64.py 8192 2 > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/final/pmc_cfg5_f64/write -- python3 $R/tools/bench_wow64.py 8192 2 > /dev/null 2>&1
unset WT_NO_WOW_OVERLAP
bash $R/tools/pmc_cfg5.sh gpurun_out/final/pmc_cfg5_sq > /dev/null 2>&1
cd $R; cat gpurun_out/final/pytest_gpu.log 2>/dev/null | tail -2; cut -c1-300 gpurun_out/final/bench.json
# keep only the small summary files of the rocprof runs (the merge back is capped at 64 MiB)
find gpurun_out/final -name "*_kernel_trace.csv" -size +2M -delete
find gpurun_out/final -name "*.db" -delete 2>/dev/null
du -sh gpurun_out/final
