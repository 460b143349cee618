#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...] : hipcc build of libwatroo_hip.so into variants/NAME.so
# (A/B libraries for tools/try_variants.sh; select one with WATROO_HIP_LIB=variants/NAME.so)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p variants
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -shared -Wall -Wno-unused-function "$@" \
    -o variants/$name.so wavelets_amd/csrc/wt_api.hip -ldl
echo built variants/$name.so
