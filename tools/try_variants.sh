#!/bin/bash
# tuning helper: run bench.py --brief against the default library and every library in variants/
# (WATROO_HIP_LIB), interleaved and repeated so that clock drift averages out.
# Extra environment per run: lines "NAME ENV=VAL ..." in the file given as $VARIANT_ENVS.
cd "$(dirname "$0")/.."
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -1
REPS=${REPS:-3}
for rep in $(seq $REPS); do
  echo -n "default: "; python bench.py --brief --steps 30 --no-build "$@"
  if [ -n "$VARIANT_ENVS" ]; then
    while read -r name envs; do
      [ -z "$name" ] && continue
      echo -n "$name: "; env $envs python bench.py --brief --steps 30 --no-build "$@"
    done < "$VARIANT_ENVS"
  fi
  for lib in variants/*.so; do
    [ -e "$lib" ] || continue
    echo -n "$(basename $lib .so): "
    WATROO_HIP_LIB=$PWD/$lib python bench.py --brief --steps 30 --no-build "$@"
  done
done
