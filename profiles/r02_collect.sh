#!/bin/bash
# Copies the judged summaries of a profiles/r02_final.sh run from gpurun_out/final (scratch) into profiles/r02/final and
# refreshes profiles/r02/pmc_p{1,2}_s1.json (read by bench.py for roofline.traffic).
set -e
cd "$(dirname "$0")/.."
SRC=gpurun_out/final; DST=profiles/r02/final
mkdir -p $DST
cp $SRC/bench_p1_s1.json $SRC/bench_p1_s1_driver_flags.json $SRC/bench_p2_s1.json $SRC/bench_p1_s32.json $SRC/bench_configs.json $SRC/split_stages.json $DST/
for c in enum_planar enum_rows enum_noafter_planar observe step_auto_1p step_auto_2p loop_1p step_obs_1p loop_2p step_obs_2p; do cp $SRC/kernel_$c.json $DST/; done
cp $SRC/pytest_gpu.log $DST/pytest_gpu.txt
cp $SRC/prof_p1.log $DST/bench_under_profiler_p1_s1.txt
cp $SRC/prof_p1_unchained.log $DST/bench_under_profiler_p1_s1_unchained.txt
[ -f $SRC/order_check.txt ] && cp $SRC/order_check.txt profiles/r02/order_check.txt
newest() { ls -t $(find "$1" -name "$2") | head -1; }
cp "$(newest $SRC/prof_p1 '*kernel_stats.csv')" $DST/kernel_stats_p1_s1.csv
cp "$(newest $SRC/prof_p2 '*kernel_stats.csv')" $DST/kernel_stats_p2_s1.csv
cp "$(newest $SRC/prof_p1_unchained '*kernel_stats.csv')" $DST/kernel_stats_p1_s1_unchained.csv
cp "$(newest $SRC/prof_p2_unchained '*kernel_stats.csv')" $DST/kernel_stats_p2_s1_unchained.csv
for c in enum_planar enum_rows enum_noafter_planar observe step_auto_1p step_auto_2p loop_1p step_obs_1p loop_2p step_obs_2p split; do cp "$(newest $SRC/prof_$c '*kernel_stats.csv')" $DST/kernel_stats_$c.csv; done
for t in p1 p2 calib; do python profiles/pmc_summary.py $SRC/pmc_$t > /dev/null; done
cp $SRC/pmc_p1/summary.json $DST/pmc_p1_s1_summary.json
cp $SRC/pmc_p2/summary.json $DST/pmc_p2_s1_summary.json
cp $SRC/pmc_calib/summary.json $DST/pmc_calib_s0_summary.json
python profiles/make_traffic_json.py $DST/pmc_p1_s1_summary.json 1 1 profiles/r02/pmc_p1_s1.json k_chain
python profiles/make_traffic_json.py $DST/pmc_p2_s1_summary.json 2 1 profiles/r02/pmc_p2_s1.json "k_duo<6"
python profiles/make_traffic_json.py $DST/pmc_calib_s0_summary.json 1 0 profiles/r02/pmc_calib_s0.json k_chain
head -3 $DST/kernel_stats_p1_s1.csv
