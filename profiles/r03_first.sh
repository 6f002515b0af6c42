#!/bin/bash
# Round 3, first GPU call: parity suite (with the new give-up tests), bench lines, the chained period from rocprofv3's own
# dispatch timestamps (pre-queued run), PMC traffic with SERIALISED dispatches (+ calibration, + the overlapped run for comparison).
set -x
set -e -o pipefail
O=gpurun_out/r03a
mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_chain_fallback.py -m gpu -q -x > $O/pytest_fallback.log 2>&1 || { tail -40 $O/pytest_fallback.log; exit 1; }
tail -1 $O/pytest_fallback.log
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -1 $O/pytest_gpu.log
timeout -k 10 300 python bench.py > $O/bench_p1_s1.json 2> $O/bench_p1_s1.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_p1_s1_driver_flags.json 2>/dev/null
cut -c1-300 $O/bench_p1_s1.json
prof() { (cd /tmp; export TMPDIR=/tmp; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/$1 -- python3 "${@:2}" > $R/$O/$1.log 2>&1); }
prof prof_p1 $R/bench.py --cpu-seconds 0
python profiles/chain_period_from_trace.py $O/prof_p1 $O/chain_period_from_trace.json > /dev/null
grep -h period_us $O/chain_period_from_trace.json | head -8
(cd /tmp; rocprofv3 -L > $R/$O/counters_list.txt 2>&1 || true)
export TETRIS_CHAIN_DEPTH=1
profiles/pmc_passes.sh $O/pmc_p1_serial mem bench.py --cpu-seconds 0 --steps 256 --warmup 16 --precondition-ms 0 --no-gpu-paced > $O/pmc_p1_serial.txt 2>&1
profiles/pmc_passes.sh $O/pmc_calib_serial mem profiles/calib.py 1 > $O/pmc_calib_serial.txt 2>&1
unset TETRIS_CHAIN_DEPTH
profiles/pmc_passes.sh $O/pmc_p1_overlapped mem bench.py --cpu-seconds 0 --steps 256 --warmup 16 --precondition-ms 0 --no-gpu-paced > $O/pmc_p1_overlapped.txt 2>&1
export TETRIS_NO_CHAIN=1
profiles/pmc_passes.sh $O/pmc_p1_unchained mem bench.py --cpu-seconds 0 --steps 256 --warmup 16 --precondition-ms 0 --no-gpu-paced > $O/pmc_p1_unchained.txt 2>&1
unset TETRIS_NO_CHAIN
python profiles/make_traffic_json.py $O/pmc_p1_serial/summary.json 1 1 $O/pmc_p1_s1.json "k_chain<1>" $O/pmc_calib_serial/summary.json $O/pmc_p1_overlapped/summary.json || true
tail -3 $O/pmc_p1_serial.txt
