# cfg5 (float32) with library variants on ONE box: bash tools/ab_cfg5.sh variants/a.so variants/b.so ...
for rep in 1 2 3; do for l in default "$@"; do
  if [ $l = default ]; then unset WATROO_HIP_LIB; else export WATROO_HIP_LIB=$PWD/$l; fi
  echo "$(basename $l): $(python bench.py --config cfg5 --no-cpu --brief --steps 10 | cut -c1-140)"
done; done
