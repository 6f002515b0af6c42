#!/bin/bash
# Everything the round's numbers come from, in one GPU call: parity suite, smoke, the bench line (with CPU baseline) at the
# default and at the driver's flags, 2-player / fused variants, secondary kernels, rocprofv3 kernel-trace summaries of the same
# bench command (chained and un-chained), PMC passes (traffic + instruction counters), calibration of the traffic counters.
set -x
set -e -o pipefail
O=gpurun_out/final
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -1 $O/pytest_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 300 python bench.py > $O/bench_p1_s1.json 2> $O/bench_p1_s1.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_p1_s1_driver_flags.json 2>/dev/null
timeout -k 10 200 python bench.py --players 2 --cpu-seconds 0 > $O/bench_p2_s1.json 2>/dev/null
timeout -k 10 200 python bench.py --steps 256 --warmup 8 --steps-per-launch 32 --cpu-seconds 0 > $O/bench_p1_s32.json 2>/dev/null
timeout -k 10 400 python profiles/bench_configs.py > $O/bench_configs.json 2> $O/bench_configs.err
R=$GRAFT_REPO_ROOT
prof() { (cd /tmp; export TMPDIR=/tmp; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/$1 -- python3 "${@:2}" > $R/$O/$1.log 2>&1); }
prof prof_p1 $R/bench.py --cpu-seconds 0
prof prof_p2 $R/bench.py --cpu-seconds 0 --players 2
# one stream, launches replayed from HIP graphs (TETRIS_GRAPH=1): under the profiler a plain launch costs the host ~8 us, more
# than the kernel takes; from a graph the kernels of the profiled run are back to back like those of the un-profiled one
export TETRIS_NO_CHAIN=1 TETRIS_GRAPH=1
prof prof_p1_unchained $R/bench.py --cpu-seconds 0
prof prof_p2_unchained $R/bench.py --cpu-seconds 0 --players 2
unset TETRIS_NO_CHAIN TETRIS_GRAPH
for c in enum_planar enum_rows enum_noafter_planar observe step_auto_1p step_auto_2p loop_1p step_obs_1p loop_2p step_obs_2p; do
  timeout -k 10 200 python profiles/kernel_prof.py $c > $O/kernel_$c.json 2>/dev/null
  prof prof_$c $R/profiles/kernel_prof.py $c
done
python profiles/order_check.py 1 > $O/order_check.txt 2>&1
timeout -k 10 200 python profiles/split_stages.py > $O/split_stages.json 2>/dev/null
prof prof_split $R/profiles/split_stages.py 256
profiles/pmc_passes.sh $O/pmc_p1 all bench.py --cpu-seconds 0 --steps 256 --warmup 16 > $O/pmc_p1.txt 2>&1
profiles/pmc_passes.sh $O/pmc_p2 mem bench.py --cpu-seconds 0 --steps 256 --warmup 16 --players 2 > $O/pmc_p2.txt 2>&1
profiles/pmc_passes.sh $O/pmc_calib mem profiles/calib.py 1 > $O/pmc_calib.txt 2>&1
cut -c1-400 $O/bench_p1_s1.json
find $O -name "*kernel_stats.csv" | head -20
