#!/bin/bash
# Round 3, second GPU call: parity suite, two-player chained vs un-chained (GPU-paced and host-paced, alternating processes),
# split-mode stage times (two-kernel vs three-kernel form), drop-in Python API throughput, two-player bench line.
set -x
set -e -o pipefail
O=gpurun_out/${OUT:-r03b}
mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -1 $O/pytest_gpu.log
for rep in 1 2; do
  timeout -k 10 120 python profiles/prequeue.py 2 >> $O/chain2_ab.txt 2>&1
  TETRIS_NO_CHAIN=1 timeout -k 10 120 python profiles/prequeue.py 2 >> $O/chain2_ab.txt 2>&1
done
cat $O/chain2_ab.txt
timeout -k 10 200 python bench.py --players 2 --cpu-seconds 0 > $O/bench_p2_s1.json 2>$O/bench_p2_s1.err
cut -c1-200 $O/bench_p2_s1.json
timeout -k 10 200 python profiles/split_stages.py > $O/split_stages.json 2>$O/split_stages.err
cat $O/split_stages.json
prof() { (cd /tmp; export TMPDIR=/tmp; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/$1 -- python3 "${@:2}" > $R/$O/$1.log 2>&1); }
prof prof_split $R/profiles/split_stages.py 256
grep k_split $O/prof_split/*/*kernel_stats.csv
timeout -k 10 300 python profiles/dropin_api.py > $O/dropin_api.json 2>$O/dropin_api.err
python -c "
import json; d=json.load(open('$O/dropin_api.json'))
for r in d['rows']: print(r['n_envs'], r['variant'], '%.3g env-steps/s' % r['env_steps_per_s'], '%.3f ms/iter' % r['ms_per_iteration'])"
for c in step_auto_1p step_auto_2p step_obs_1p step_obs_2p; do timeout -k 10 200 python profiles/kernel_prof.py $c > $O/kernel_$c.json 2>/dev/null; cut -c1-160 $O/kernel_$c.json; done
timeout -k 10 300 python profiles/dropin_breakdown.py default 4096 > $O/dropin_breakdown.json 2>/dev/null; cat $O/dropin_breakdown.json
TETRIS_NO_CHAIN=1 timeout -k 10 200 python bench.py --players 2 --cpu-seconds 0 > $O/bench_p2_s1_unchained.json 2>/dev/null; cut -c1-200 $O/bench_p2_s1_unchained.json
for rep in 1 2; do for P in 2 1; do
  timeout -k 10 120 python profiles/ab_old_lib.py profiles/_ab/libtetris_r02.so $P 2>/dev/null >> $O/ab_r02.txt
  timeout -k 10 120 python profiles/ab_old_lib.py default $P 2>/dev/null >> $O/ab_r02.txt
done; done; cat $O/ab_r02.txt
