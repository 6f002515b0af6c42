#!/bin/bash
# ASan + UBSan over the two CPU builds (GPU sanitizers are not available on the pool): the oracle (oracle/liboracle_asan.so)
# and the g++ build of the product's kernel bodies (tests/cpu_harness).  Replays golden traces incl. get_actions and colour
# planes and runs the batched paths; any report aborts.   usage: tests/sanitizers/run.sh
set -e
cd "$(dirname "$0")/../.."
make -C oracle liboracle_asan.so
g++ -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer \
    -o tests/cpu_harness/libtetris_cpu_harness_asan.so tests/cpu_harness/harness.cpp
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) python3 tests/sanitizers/asan_run.py
