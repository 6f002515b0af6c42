#!/bin/bash
# Round 3, final collection, part C: PMC passes on the XCD-affine kernels (direct dispatch forced on under the profiler:
# TETRIS_DIRECT_UNDER_TOOLS=1).  rocprofv3 serialises the dispatches of a --pmc run and its own packets stand between them, so what
# is counted is the affine kernel with whatever the tool's packets leave in the L2s — an upper bound for the fabric traffic of the
# un-profiled run, where nothing stands between a queue's launches.
set -x
O=gpurun_out/final3c
mkdir -p $O
ARGS="--cpu-seconds 0 --steps 256 --warmup 16 --precondition-ms 0 --no-gpu-paced"
export TETRIS_DIRECT_UNDER_TOOLS=1
profiles/pmc_passes.sh $O/pmc_p1_affine mem bench.py $ARGS > $O/pmc_p1_affine.txt 2>&1
tail -3 $O/pmc_p1_affine.txt
