#!/bin/bash
# TEST INFRASTRUCTURE: builds the CPU driver-logic checker (see cpu_prims.cpp) into oracle/_build/.
set -e
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
mkdir -p "$root/oracle/_build"
g++ -O2 -fPIC -shared -std=c++17 -o "$root/oracle/_build/libhostcheck.so" \
    "$here/cpu_prims.cpp" "$root/bayesian_ensembling_amd/csrc/driver.cpp"
echo "built $root/oracle/_build/libhostcheck.so"
