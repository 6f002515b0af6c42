#!/bin/bash
# first runs of the direct-dispatch path: small soaks against the oracle (P = 1, 2), then timing lines
set -e
mkdir -p gpurun_out/direct
export TETRIS_TIMING=1
timeout -k 10 200 python tests/tools/chain_soak.py 300 1 > gpurun_out/direct/soak_p1.txt 2>&1 || { tail -20 gpurun_out/direct/soak_p1.txt; exit 1; }
tail -4 gpurun_out/direct/soak_p1.txt
timeout -k 10 200 python tests/tools/chain_soak.py 300 2 > gpurun_out/direct/soak_p2.txt 2>&1 || { tail -20 gpurun_out/direct/soak_p2.txt; exit 1; }
tail -4 gpurun_out/direct/soak_p2.txt
for d in 1 0; do
  TETRIS_DIRECT=$d timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 > gpurun_out/direct/bench20_d$d.json 2> gpurun_out/direct/bench20_d$d.err || { tail -20 gpurun_out/direct/bench20_d$d.err; exit 1; }
  python -c "import json,sys; d=json.loads(open('gpurun_out/direct/bench20_d$d.json').read().strip().splitlines()[-1]); print('direct=$d K=20:', d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['frac'], d['roofline'].get('launch_us_events'))"
  grep "tetris timing" gpurun_out/direct/bench20_d$d.err | tail -3
  TETRIS_DIRECT=$d timeout -k 10 200 python bench.py --cpu-seconds 0 > gpurun_out/direct/bench_d$d.json 2> gpurun_out/direct/bench_d$d.err || { tail -20 gpurun_out/direct/bench_d$d.err; exit 1; }
  python -c "import json,sys; d=json.loads(open('gpurun_out/direct/bench_d$d.json').read().strip().splitlines()[-1]); print('direct=$d default K:', d['value']/1e9, d['ms_per_step']*1e3, d['roofline']['frac'], d['roofline'].get('launch_us_events'))"
  grep "tetris timing" gpurun_out/direct/bench_d$d.err | tail -2
done
