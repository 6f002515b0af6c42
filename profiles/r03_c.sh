#!/bin/bash
set -x
set -e -o pipefail
O=gpurun_out/${OUT:-r03f}
mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -1 $O/pytest_gpu.log
for rep in 1 2; do
  timeout -k 10 120 python profiles/ab_old_lib.py profiles/_ab/libtetris_r02.so 2 2>/dev/null >> $O/ab_r02.txt
  timeout -k 10 120 python profiles/ab_old_lib.py default 2 2>/dev/null >> $O/ab_r02.txt
done; cat $O/ab_r02.txt
for N in 16384 28672; do
  timeout -k 10 120 python profiles/prequeue.py 2 default $N 2>/dev/null >> $O/chain2_sizes.txt
  TETRIS_NO_CHAIN=1 timeout -k 10 120 python profiles/prequeue.py 2 default $N 2>/dev/null >> $O/chain2_sizes.txt
done; cat $O/chain2_sizes.txt
for i in 1 2 3; do TETRIS_TIMING=1 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 2>> $O/timing20.txt | cut -c1-160; done
grep "20 launches" $O/timing20.txt
for c in step_auto_2p step_obs_2p; do timeout -k 10 200 python profiles/kernel_prof.py $c > $O/kernel_$c.json 2>/dev/null; cut -c1-160 $O/kernel_$c.json; done
