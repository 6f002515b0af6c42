#!/bin/bash
# quick GPU check used while tuning: parity tests, then the three bench shapes (no CPU leg)
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/pytest_gpu.log 2>&1 || { tail -20 gpurun_out/pytest_gpu.log; exit 1; }
tail -1 gpurun_out/pytest_gpu.log
timeout -k 10 300 python bench.py --steps 2048 --warmup 64 --cpu-seconds 0 > gpurun_out/bench1.log 2>&1 &&
timeout -k 10 200 python bench.py --steps 2048 --warmup 64 --players 2 --cpu-seconds 0 > gpurun_out/bench_p2.log 2>&1 &&
timeout -k 10 200 python bench.py --steps 256 --warmup 8 --steps-per-launch 32 --cpu-seconds 0 > gpurun_out/bench_fused.log 2>&1
python - <<PY
import json
for f in ("bench1","bench_p2","bench_fused"):
    try:
        d=json.loads(open(f"gpurun_out/{f}.log").read().strip().splitlines()[-1])
        print(f, "%.2f G/s"%(d["value"]/1e9), "%.2f us"%d["roofline"]["launch_us"], "frac %.3f"%d["roofline"]["frac"])
    except Exception as e:
        print(f, "FAILED", e); print(open(f"gpurun_out/{f}.log").read()[-800:])
PY
