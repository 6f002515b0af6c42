#!/bin/bash
# alternating-process A/B of library builds, GPU-paced (profiles/prequeue.py): profiles/ab_libs.sh P lib1 lib2 ...   (3 rounds)
P=$1; shift
for round in 1 2 3; do
  for lib in "$@"; do timeout -k 10 120 python profiles/prequeue.py $P $lib 2>&1 | tail -1; done
done
