#!/bin/bash
# Everything the round's numbers come from, in one GPU call: parity suite, smoke, the bench line (with CPU baseline), the
# 2-player / fused variants, the rocprofv3 kernel-trace summary of the same bench command, PMC traffic passes.
set -x
set -e -o pipefail      # a failed or timed-out GPU step ends the call: no further GPU step is started after it
mkdir -p gpurun_out/final
timeout -k 10 400 python -m pytest tests -m gpu -q > gpurun_out/final/pytest_gpu.log 2>&1; tail -1 gpurun_out/final/pytest_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; tail -1 gpurun_out/final/smoke.log
timeout -k 10 300 python bench.py > gpurun_out/final/bench_p1_s1.json 2> gpurun_out/final/bench_p1_s1.err
timeout -k 10 200 python bench.py --players 2 --cpu-seconds 0 > gpurun_out/final/bench_p2_s1.json 2>/dev/null
timeout -k 10 200 python bench.py --steps 256 --warmup 8 --steps-per-launch 32 --cpu-seconds 0 > gpurun_out/final/bench_p1_s32.json 2>/dev/null
timeout -k 10 300 python profiles/bench_configs.py > gpurun_out/final/bench_configs.json 2>/dev/null
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_p1 -- python3 $R/bench.py --cpu-seconds 0 > $R/gpurun_out/final/prof_p1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_p2 -- python3 $R/bench.py --cpu-seconds 0 --players 2 > $R/gpurun_out/final/prof_p2.log 2>&1
cd $R
profiles/pmc_passes.sh gpurun_out/final/pmc_p1 all bench.py --cpu-seconds 0 --steps 256 --warmup 16 > gpurun_out/final/pmc_p1.txt 2>&1
profiles/pmc_passes.sh gpurun_out/final/pmc_p2 mem bench.py --cpu-seconds 0 --steps 256 --warmup 16 --players 2 > gpurun_out/final/pmc_p2.txt 2>&1
profiles/pmc_passes.sh gpurun_out/final/pmc_calib mem profiles/calib.py 1 > gpurun_out/final/pmc_calib.txt 2>&1
cat gpurun_out/final/bench_p1_s1.json | cut -c1-300
find gpurun_out/final -name "*kernel_stats.csv" | head
