#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...] : build of libwatroo_hip.so into variants/NAME.so with extra
# compiler flags (A/B libraries for tools/try_variants.sh; select one with WATROO_HIP_LIB=variants/NAME.so)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p variants
python - "$name" "$@" <<'PY'
import sys
import __graft_entry__ as e
name, flags = sys.argv[1], sys.argv[2:]
e.build(force=True, extra_flags=flags, lib=f"variants/{name}.so")
print(f"built variants/{name}.so")
PY
