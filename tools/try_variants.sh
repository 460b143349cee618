#!/bin/bash
# tuning helper: run bench.py --brief against every library in variants/ (WATROO_HIP_LIB)
cd "$(dirname "$0")/.."
rocm-smi --showuniqueid 2>/dev/null | grep -i "unique" | head -1
for rep in 1 2; do
  for lib in variants/*.so; do
    echo -n "$(basename $lib .so): "
    WATROO_HIP_LIB=$PWD/$lib python bench.py --brief --steps 30 "$@"
  done
done
