# float64 cfg5 with library variants on ONE box: bash tools/ab_tab64.sh variants/a.so ...  (default = the in-tree library)
for rep in 1 2 3; do for l in default "$@"; do
  if [ $l = default ]; then unset WATROO_HIP_LIB; else export WATROO_HIP_LIB=$PWD/$l; fi
  echo "$(basename $l): $(python tools/bench_wow64.py 8192 5 | grep 'ms/step\|float64 / float32\|wt64_bilateral_kernel' | cut -c1-100 | tr '\n' ' ')"
done; done
