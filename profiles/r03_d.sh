#!/bin/bash
set -x
set -e -o pipefail
O=gpurun_out/${OUT:-r03g}
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -1 $O/pytest_gpu.log
for rep in 1 2 3; do for P in 1 2; do
  timeout -k 10 120 python profiles/prequeue.py $P profiles/_ab/libtetris_head.so 2>/dev/null >> $O/ab_head.txt
  timeout -k 10 120 python profiles/prequeue.py $P default 2>/dev/null >> $O/ab_head.txt
  timeout -k 10 120 python profiles/ab_old_lib.py profiles/_ab/libtetris_head.so $P 2>/dev/null >> $O/ab_head_unchained.txt
  timeout -k 10 120 python profiles/ab_old_lib.py default $P 2>/dev/null >> $O/ab_head_unchained.txt
done; done
cat $O/ab_head.txt $O/ab_head_unchained.txt
