#!/bin/bash
# Copies the judged summaries of a profiles/final_round.sh run from gpurun_out/final (scratch) into profiles/<round>/final
# and refreshes profiles/pmc_p{1,2}_s1.json (read by bench.py for roofline.traffic).   usage: profiles/collect_final.sh r01
set -e
cd "$(dirname "$0")/.."
R=${1:-r01}; SRC=gpurun_out/final; DST=profiles/$R/final
mkdir -p $DST
cp $SRC/bench_p1_s1.json $SRC/bench_p2_s1.json $SRC/bench_p1_s32.json $SRC/bench_configs.json $DST/
cp $SRC/pytest_gpu.log $DST/pytest_gpu.txt
newest() { ls -t $(find "$1" -name "$2") | head -1; }
cp "$(newest $SRC/prof_p1 '*kernel_stats.csv')" $DST/kernel_stats_p1_s1.csv
cp "$(newest $SRC/prof_p2 '*kernel_stats.csv')" $DST/kernel_stats_p2_s1.csv
for t in p1 p2 calib; do python profiles/pmc_summary.py $SRC/pmc_$t > /dev/null; done
cp $SRC/pmc_p1/summary.json $DST/pmc_p1_s1_summary.json
cp $SRC/pmc_p2/summary.json $DST/pmc_p2_s1_summary.json
cp $SRC/pmc_calib/summary.json $DST/pmc_calib_s0_summary.json
python profiles/make_traffic_json.py $DST/pmc_p1_s1_summary.json 1 1 profiles/pmc_p1_s1.json
python profiles/make_traffic_json.py $DST/pmc_p2_s1_summary.json 2 1 profiles/pmc_p2_s1.json
head -3 $DST/kernel_stats_p1_s1.csv
