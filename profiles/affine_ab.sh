#!/bin/bash
# XCD-affine hand-off experiment: period A/B (GPU-paced and host-paced, alternating processes) + a bit-exact soak of the variant
set -e
mkdir -p gpurun_out/affine
V=${1:-profiles/_build/libtetris_affine0.so}
bash profiles/ab_libs.sh 1 default $V > gpurun_out/affine/ab_p1.txt 2>&1
cat gpurun_out/affine/ab_p1.txt
timeout -k 10 300 python tests/tools/chain_soak.py 2000 1 $V > gpurun_out/affine/soak_p1.txt 2>&1
tail -3 gpurun_out/affine/soak_p1.txt
