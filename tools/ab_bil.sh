# bilateral-kernel work, one box: parity of the bilateral paths, then cfg5 (float32) A/B of library variants
# with the per-kernel breakdown.  bash tools/ab_bil.sh variants/a.so ...
mkdir -p gpurun_out/bil
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "bilateral or wow or cfg5 or inline_variance" > gpurun_out/bil/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 gpurun_out/bil/pytest.log
[ $rc = 0 ] || exit $rc
for rep in 1 2 3; do for l in default "$@"; do
  if [ $l = default ]; then unset WATROO_HIP_LIB; else export WATROO_HIP_LIB=$PWD/$l; fi
  echo "$(basename $l): $(python bench.py --config cfg5 --no-cpu --brief --steps 10 | cut -c1-160)"
done; done
unset WATROO_HIP_LIB
python tools/bench_wow64.py 8192 5 > gpurun_out/bil/wow64_default.txt 2>&1; grep -i "bilateral\|cfg5\|float32" gpurun_out/bil/wow64_default.txt | head -8
