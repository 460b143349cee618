#!/bin/bash
# The C oracle under AddressSanitizer + UBSan on the CPU (the GPU pool runs no sanitizers): builds
# oracle/atrous_ref.c with -fsanitize=address,undefined into a temporary library and runs the oracle
# test files against it.  The checker must not have memory errors of its own.
set -e
cd "$(dirname "$0")/.."
D=$(mktemp -d /tmp/wt_asan_XXXX)
gcc -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -march=x86-64-v3 -fopenmp -fPIC -ffp-contract=off \
    -fno-fast-math -Wall -shared -o $D/liboracle.so oracle/atrous_ref.c -lm
WT_ORACLE_LIB=$D/liboracle.so LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
    ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_oracle_c.py tests/test_oracle_golden.py -x -q "$@"
