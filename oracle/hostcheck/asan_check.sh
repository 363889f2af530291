#!/bin/bash
# TEST INFRASTRUCTURE: the host-side driver (csrc/driver.cpp: blocked Cholesky with look-ahead, recursive solves,
# padding / bordered-row paths, loss sampling) under AddressSanitizer + UBSan on the CPU primitives, through the
# same pytest cases (tests/test_hostcheck_driver.py).  GPU sanitizers are not available on the pool; this covers the
# sequencing and indexing logic that decides which device addresses the kernels are given.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
lib="$root/oracle/_build/libhostcheck.so"
"$here/build.sh" > /dev/null
cp "$lib" /tmp/libhostcheck_plain.so
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer \
    -o "$lib" "$here/cpu_prims.cpp" "$root/bayesian_ensembling_amd/csrc/driver.cpp"
trap 'cp /tmp/libhostcheck_plain.so "$lib"' EXIT
cd "$root"
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_hostcheck_driver.py -x -q -p no:cacheprovider
