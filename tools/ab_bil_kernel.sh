# device time of the float32 bilateral march by dilation (tools/bench_bil.py) for library variants on ONE box, two rounds
# bash tools/ab_bil_kernel.sh variants/a.so ...   (the in-tree library is always first)
for rep in 1 2; do for l in default "$@"; do
  if [ $l = default ]; then unset WATROO_HIP_LIB; else export WATROO_HIP_LIB=$PWD/$l; fi
  echo "$(basename $l): $(python tools/bench_bil.py 8192 5 | tr -s ' ' | awk '/d=/{printf "%s ", $3} /per transform/{print "| " $5 " ms per transform"}')"
done; done
