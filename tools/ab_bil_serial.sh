# serial order (no side stream): clean per-kernel time of wt_bilateral2_kernel in the cfg5 flow for library variants,
# one box.  bash tools/ab_bil_serial.sh variants/a.so ...   (the in-tree library is always the first column)
export WT_NO_WOW_OVERLAP=1
for rep in 1 2; do for l in default "$@"; do
  if [ $l = default ]; then unset WATROO_HIP_LIB; else export WATROO_HIP_LIB=$PWD/$l; fi
  echo "$(basename $l): $(python tools/bench_wow64.py 8192 5 f32only 2>&1 | grep -E 'ms/step|bilateral2' | head -2 | tr -s ' ' | tr '\n' ' ')"
done; done
