#!/bin/bash
# experiments of round 2 on ONE box: state layout (tiles / rows) x launch ordering (chained / one stream), bench.py's own clock
set -e -o pipefail
O=gpurun_out/${OUT:-r02x}
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_full_size.py -m gpu -x -q > $O/pytest_full.log 2>&1 || { tail -30 $O/pytest_full.log; exit 1; }
tail -1 $O/pytest_full.log
for rep in 1 2; do
for lay in tiles rows; do
for v in chain nochain; do
  if [ $v = nochain ]; then export TETRIS_NO_CHAIN=1; else unset TETRIS_NO_CHAIN; fi
  if [ $lay = rows ]; then export BENCH_LIB_PATH=$PWD/drl-tetris_amd/lib/exp_rows.so; else unset BENCH_LIB_PATH; fi
  timeout -k 10 300 python bench.py --cpu-seconds 0 > $O/bench_${lay}_$v.json 2> $O/bench_${lay}_$v.err || { tail $O/bench_${lay}_$v.err; exit 1; }
  timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 20 --warmup 5 > $O/bench_${lay}_${v}_20.json 2>/dev/null
  timeout -k 10 300 python bench.py --cpu-seconds 0 --players 2 > $O/bench_${lay}_${v}_p2.json 2>/dev/null
  python - <<PY
import json
for f in ("bench_${lay}_$v","bench_${lay}_${v}_20","bench_${lay}_${v}_p2"):
    d=json.loads(open(f"$O/{f}.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("rep$rep", f, "%.2f G/s"%(d["value"]/1e9), "wall %.2f us"%r["launch_us"], "events %.2f"%r["launch_us_events"], "frac %.3f"%r["frac"])
PY
done; done; done
unset TETRIS_NO_CHAIN BENCH_LIB_PATH
