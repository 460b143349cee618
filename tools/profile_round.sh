#!/bin/bash
# End-of-session evidence: GPU tests, bench lines (headline + cfg2/3/5), rocprofv3 kernel stats and
# the FETCH_SIZE / WRITE_SIZE passes.  Results under gpurun_out/final; copy the summaries to
# profiles/ with tools/pmc_summary.py.  The profiled program sits directly behind `--` and is
# told not to build (no child process under the profiler's preload).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
rm -rf gpurun_out/final; mkdir -p gpurun_out/final
python -c "import __graft_entry__ as e; e.build()"
timeout -k 10 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -5 > gpurun_out/final/pytest_gpu.log
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
for c in cfg2 cfg3 cfg5; do
  timeout -k 10 600 python bench.py --config $c --no-build > gpurun_out/final/bench_$c.json 2> gpurun_out/final/bench_$c.err
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof -- python3 $R/bench.py --steps 20 --no-cpu --no-build > $R/gpurun_out/final/rocprof_bench.log 2>&1
for c in cfg2 cfg3 cfg5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_$c -- python3 $R/bench.py --config $c --steps 5 --no-cpu --no-build > $R/gpurun_out/final/rocprof_$c.log 2>&1
done
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/final/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-build --brief > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/final/pmc_write -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-build --brief > /dev/null 2>&1
cd $R; cat gpurun_out/final/pytest_gpu.log | tail -2; cut -c1-300 gpurun_out/final/bench.json
