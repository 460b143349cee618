rm -rf gpurun_out/final; mkdir -p gpurun_out/final
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -5 > gpurun_out/final/pytest_gpu.log
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
python tools/bench_configs.py 2 3 5 w > gpurun_out/final/configs.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof -- python3 $R/bench.py --steps 20 --no-cpu > $R/gpurun_out/final/rocprof_bench.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/final/pmc_fetch -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --brief > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/final/pmc_write -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --brief > /dev/null 2>&1
cd $R; cat gpurun_out/final/pytest_gpu.log | tail -2; cut -c1-300 gpurun_out/final/bench.json
