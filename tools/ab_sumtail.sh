# cfg5 (float32 and float64, 8192^2) with the early part of the plane sum off (tail = 99 planes: never early) and
# on with 3 / 4 / 5 planes left for the end, interleaved on ONE box:  bash tools/ab_sumtail.sh
for rep in 1 2 3; do for t in 99 3 4 5; do
  export WATROO_HIP_SUM_TAIL=$t
  echo "tail=$t: $(python tools/bench_wow64.py 8192 5 | grep 'ms/step\|float64 / float32' | tr '\n' ' ')"
done; done
