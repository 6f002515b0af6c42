#!/bin/bash
# same-box experiment: games per wave of the chained kernel (64 / 32 / 16)
set -e -o pipefail
O=gpurun_out/${OUT:-r02z}
mkdir -p $O
for l in 64 32 16; do
  if [ $l = 64 ]; then unset TETRIS_LIB BENCH_LIB_PATH; else export TETRIS_LIB=$PWD/drl-tetris_amd/lib/exp_chain$l.so BENCH_LIB_PATH=$PWD/drl-tetris_amd/lib/exp_chain$l.so; fi
  timeout -k 10 300 python tests/tools/chain_parity.py 2>&1 | tail -1
  for rep in 1 2; do
  timeout -k 10 300 python bench.py --cpu-seconds 0 > $O/bench_l$l.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
  timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 20 --warmup 5 > $O/bench_l${l}_20.json 2>/dev/null
  python - <<PY
import json
for f in ("bench_l$l","bench_l${l}_20"):
    d=json.loads(open(f"$O/{f}.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("lanes $l", f, "%.2f G/s"%(d["value"]/1e9), "wall %.2f us"%r["launch_us"], "frac %.3f"%r["frac"], "unchained", round(r["unchained"]["launch_us"],2))
PY
  done
done
