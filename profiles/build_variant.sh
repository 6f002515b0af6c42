#!/bin/bash
# profiles/build_variant.sh NAME -DFLAG...  ->  profiles/_build/libtetris_NAME.so (experiment builds of the library; not shipped)
set -e
name=$1; shift
mkdir -p profiles/_build
for u in tetris_hip tetris_hip_multi; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC "$@" -c -o profiles/_build/${u}_$name.o drl-tetris_amd/csrc/$u.hip &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o profiles/_build/libtetris_$name.so profiles/_build/tetris_hip_$name.o profiles/_build/tetris_hip_multi_$name.o -L/opt/rocm/lib -lhsa-runtime64
rm -f profiles/_build/*_$name.o
ls -la profiles/_build/libtetris_$name.so
