#!/bin/bash
# quick same-box numbers while tuning: bench (1p, 2p, fused) + secondary configs; optional parity subset first (PYTEST_K)
set -e -o pipefail
O=gpurun_out/${OUT:-r02q}
mkdir -p $O
if [ -n "$PYTEST_K" ]; then timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "$PYTEST_K" > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }; tail -1 $O/pytest_gpu.log; fi
timeout -k 10 300 python bench.py --cpu-seconds 0 > $O/bench_p1_s1.json 2> $O/bench_p1_s1.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > $O/bench_driver_flags.json 2>/dev/null
timeout -k 10 200 python bench.py --players 2 --cpu-seconds 0 > $O/bench_p2_s1.json 2>/dev/null
timeout -k 10 200 python bench.py --steps 256 --warmup 8 --steps-per-launch 32 --cpu-seconds 0 > $O/bench_p1_s32.json 2>/dev/null
if [ -z "$NO_CONFIGS" ]; then timeout -k 10 400 python profiles/bench_configs.py > $O/bench_configs.json 2> $O/bench_configs.err || { tail -20 $O/bench_configs.err; exit 1; }; fi
python - <<PY
import json
for f in ("bench_p1_s1","bench_driver_flags","bench_p2_s1","bench_p1_s32"):
    d=json.loads(open(f"$O/{f}.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print(f, "%.2f G/s"%(d["value"]/1e9), "wall %.2f us"%r["launch_us"], "events", r["launch_us_events"], "frac", r["frac"])
try:
    d=json.load(open("$O/bench_configs.json"))
    for k,v in d.items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ("us_per_call","us_per_launch","frac_of_8TBps","env_steps_per_s")})
except Exception as e: print("no configs", e)
PY
