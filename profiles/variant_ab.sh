#!/bin/bash
# profiles/variant_ab.sh NAME [P]: period A/B of profiles/_build/libtetris_NAME.so against the in-tree build (GPU-paced and host-paced,
# alternating processes), then a bit-exact soak of the variant
set -e
name=$1; P=${2:-1}
mkdir -p gpurun_out/variant
V=profiles/_build/libtetris_$name.so
bash profiles/ab_libs.sh $P default $V > gpurun_out/variant/ab_${name}_p$P.txt 2>&1
cat gpurun_out/variant/ab_${name}_p$P.txt
timeout -k 10 300 python tests/tools/chain_soak.py 2000 $P $V > gpurun_out/variant/soak_${name}_p$P.txt 2>&1
tail -3 gpurun_out/variant/soak_${name}_p$P.txt
