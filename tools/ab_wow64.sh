# cfg5 in float64 with library variants on ONE box: bash tools/ab_wow64.sh variants/a.so ...
export WT_NO_WOW_OVERLAP=1
for rep in 1 2; do for l in default "$@"; do
  if [ $l = default ]; then unset WATROO_HIP_LIB; else export WATROO_HIP_LIB=$PWD/$l; fi
  echo "$(basename $l): $(python tools/bench_wow64.py 8192 3 2>&1 | grep 'float64 bilateral=True\|wt64_bilateral_kernel' | tr '\n' ' ' | cut -c1-220)"
done; done
