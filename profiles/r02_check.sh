#!/bin/bash
# round-2 GPU check: parity suite, smoke, the bench line at the driver's flags and at the default, kernel-trace of the same command
set -x
set -e -o pipefail
O=gpurun_out/r02a
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -3 $O/pytest_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_driver_flags.json 2> $O/bench_driver_flags.err
timeout -k 10 300 python bench.py --cpu-seconds 0 > $O/bench_p1_s1.json 2> $O/bench_p1_s1.err
timeout -k 10 200 python bench.py --players 2 --cpu-seconds 0 > $O/bench_p2_s1.json 2>/dev/null
timeout -k 10 200 python bench.py --steps 256 --warmup 8 --steps-per-launch 32 --cpu-seconds 0 > $O/bench_p1_s32.json 2>/dev/null
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_p1 -- python3 $R/bench.py --cpu-seconds 0 > $R/$O/prof_p1.log 2>&1
cd $R
python - <<PY
import json
for f in ("bench_driver_flags","bench_p1_s1","bench_p2_s1","bench_p1_s32"):
    d=json.loads(open(f"$O/{f}.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print(f, "%.2f G/s"%(d["value"]/1e9), "wall %.2f us"%r["launch_us"], "events", r["launch_us_events"], "frac", r["frac"], "frac_kernel", r["frac_kernel"], d["device"])
PY
find $O -name "*kernel_stats.csv" | head -3
