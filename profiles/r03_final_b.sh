#!/bin/bash
# Round 3, final collection, part B: PMC passes.  Traffic of the chained kernels from raw TCC request counters (one- and
# two-player), the zero-step calibration of the read factor, the un-chained kernels for comparison, and the SQ instruction /
# stall counters per wave.  One counter group per pass (separate runs), counters only with --kernel-trace.
set -x
set -e -o pipefail
O=gpurun_out/final3b
mkdir -p $O
ARGS="--cpu-seconds 0 --steps 256 --warmup 16 --precondition-ms 0 --no-gpu-paced"
export TETRIS_CHAIN_DEPTH=1
profiles/pmc_passes.sh $O/pmc_p1_serial all bench.py $ARGS > $O/pmc_p1_serial.txt 2>&1
profiles/pmc_passes.sh $O/pmc_calib_p1 mem profiles/calib.py 1 > $O/pmc_calib_p1.txt 2>&1
profiles/pmc_passes.sh $O/pmc_p2_serial all bench.py $ARGS --players 2 > $O/pmc_p2_serial.txt 2>&1
profiles/pmc_passes.sh $O/pmc_calib_p2 mem profiles/calib.py 2 > $O/pmc_calib_p2.txt 2>&1
unset TETRIS_CHAIN_DEPTH
profiles/pmc_passes.sh $O/pmc_p1_three_streams mem bench.py $ARGS > $O/pmc_p1_three_streams.txt 2>&1
export TETRIS_NO_CHAIN=1
profiles/pmc_passes.sh $O/pmc_p1_unchained all bench.py $ARGS > $O/pmc_p1_unchained.txt 2>&1
profiles/pmc_passes.sh $O/pmc_p2_unchained all bench.py $ARGS --players 2 > $O/pmc_p2_unchained.txt 2>&1
unset TETRIS_NO_CHAIN
python profiles/make_traffic_json.py $O/pmc_p1_serial/summary.json 1 1 $O/pmc_p1_s1.json "k_chain<1>" $O/pmc_calib_p1/summary.json $O/pmc_p1_three_streams/summary.json
python profiles/make_traffic_json.py $O/pmc_p2_serial/summary.json 2 1 $O/pmc_p2_s1.json "k_duo<6, true" $O/pmc_calib_p2/summary.json
tail -4 $O/pmc_p1_serial.txt
