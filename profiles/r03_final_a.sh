#!/bin/bash
# Round 3, final collection, part A: parity suite, smoke, the bench lines, rocprofv3 kernel traces of the SAME bench commands
# (chained one- and two-player, un-chained from graphs), the chained periods from the traces' own timestamps, split mode, secondary
# kernels, the drop-in Python API.  Everything lands in gpurun_out/final3a; profiles/r03_collect.sh copies the judged files.
set -x
set -e -o pipefail
O=gpurun_out/final3a
mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -1 $O/pytest_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 300 python bench.py > $O/bench_p1_s1.json 2> $O/bench_p1_s1.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_p1_s1_driver_flags.json 2>/dev/null
timeout -k 10 200 python bench.py --players 2 --cpu-seconds 0 > $O/bench_p2_s1.json 2>/dev/null
# the same lines without the XCD-affine kernels (direct dispatch, write-through hand-over), then through hipLaunchKernel on streams
TETRIS_AFFINE=0 timeout -k 10 300 python bench.py --cpu-seconds 0 > $O/bench_p1_s1_write_through.json 2>/dev/null
TETRIS_AFFINE=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_p1_s1_driver_flags_write_through.json 2>/dev/null
TETRIS_AFFINE=0 timeout -k 10 200 python bench.py --players 2 --cpu-seconds 0 > $O/bench_p2_s1_write_through.json 2>/dev/null
{ timeout -k 10 300 python tests/tools/chain_soak.py 10000 1; timeout -k 10 300 python tests/tools/chain_soak.py 10000 2; } 2>&1 | grep -v amdgpu.ids > $O/chain_soak.txt
cat $O/chain_soak.txt
# the same lines with the launches going through hipLaunchKernel on streams instead of the batch's own queues
TETRIS_DIRECT=0 timeout -k 10 300 python bench.py --cpu-seconds 0 > $O/bench_p1_s1_streams.json 2>/dev/null
TETRIS_DIRECT=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_p1_s1_driver_flags_streams.json 2>/dev/null
TETRIS_DIRECT=0 timeout -k 10 200 python bench.py --players 2 --cpu-seconds 0 > $O/bench_p2_s1_streams.json 2>/dev/null
for i in 1 2 3; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver flags, direct :', round(d['value']/1e9,2), 'G', round(d['ms_per_step']*1e3,3), 'us')"; TETRIS_DIRECT=0 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('driver flags, streams:', round(d['value']/1e9,2), 'G', round(d['ms_per_step']*1e3,3), 'us')"; done > $O/driver_flags_ab.txt 2>&1
cat $O/driver_flags_ab.txt
{ timeout -k 10 200 python profiles/short_calls3.py 20 1 40; timeout -k 10 200 python profiles/short_calls3.py 20 1 40 1; timeout -k 10 200 python profiles/short_calls3.py 2048 1 6; timeout -k 10 200 python profiles/short_calls3.py 20 2 40; timeout -k 10 200 python profiles/short_calls3.py 2048 2 6; } 2>/dev/null | grep "^K=" > $O/direct_vs_streams.txt
cat $O/direct_vs_streams.txt
TETRIS_NO_CHAIN=1 timeout -k 10 200 python bench.py --cpu-seconds 0 > $O/bench_p1_s1_unchained.json 2>/dev/null
TETRIS_NO_CHAIN=1 timeout -k 10 200 python bench.py --players 2 --cpu-seconds 0 > $O/bench_p2_s1_unchained.json 2>/dev/null
timeout -k 10 200 python bench.py --steps 256 --warmup 8 --steps-per-launch 32 --cpu-seconds 0 > $O/bench_p1_s32.json 2>/dev/null
prof() { (cd /tmp; export TMPDIR=/tmp; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/$1 -- python3 "${@:2}" > $R/$O/$1.log 2>&1); }
prof prof_p1 $R/bench.py --cpu-seconds 0
prof prof_p2 $R/bench.py --cpu-seconds 0 --players 2
python profiles/chain_period_from_trace.py $O/prof_p1 $O/chain_period_from_trace_p1.json > /dev/null
# (under the profiler the launches stay on the streams: forced onto the library's queues — TETRIS_DIRECT_UNDER_TOOLS=1 — the --pmc passes of
# r03_final_c.sh went through, a --kernel-trace --stats run died inside the tool's doorbell handler: profiles/r03/direct_dispatch.txt (5))
python profiles/chain_period_from_trace.py $O/prof_p2 $O/chain_period_from_trace_p2.json > /dev/null
grep -h period_us_of $O/chain_period_from_trace_p1.json $O/chain_period_from_trace_p2.json
export TETRIS_NO_CHAIN=1 TETRIS_GRAPH=1
prof prof_p1_unchained $R/bench.py --cpu-seconds 0
prof prof_p2_unchained $R/bench.py --cpu-seconds 0 --players 2
unset TETRIS_NO_CHAIN TETRIS_GRAPH
timeout -k 10 200 python profiles/split_stages.py > $O/split_stages.json 2>/dev/null; cat $O/split_stages.json
prof prof_split $R/profiles/split_stages.py 256
for c in enum_planar observe step_auto_1p step_auto_2p step_obs_1p step_obs_2p; do
  timeout -k 10 200 python profiles/kernel_prof.py $c > $O/kernel_$c.json 2>/dev/null
  prof prof_$c $R/profiles/kernel_prof.py $c
done
timeout -k 10 300 python profiles/dropin_api.py > $O/dropin_api.json 2>/dev/null
timeout -k 10 300 python profiles/dropin_breakdown.py default 4096 > $O/dropin_breakdown.json 2>/dev/null
python profiles/order_check.py 1 > $O/order_check_p1.txt 2>&1
python profiles/order_check.py 2 > $O/order_check_p2.txt 2>&1
cut -c1-300 $O/bench_p1_s1.json
