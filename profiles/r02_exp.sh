#!/bin/bash
# experiments of round 2 on ONE box: enumerate variants (experiment builds), chained vs unchained rollout launches
set -e -o pipefail
O=gpurun_out/${OUT:-r02x}
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_full_size.py -m gpu -x -q > $O/pytest_full.log 2>&1 || { tail -30 $O/pytest_full.log; exit 1; }
tail -1 $O/pytest_full.log
for v in chain nochain; do
  if [ $v = nochain ]; then export TETRIS_NO_CHAIN=1; else unset TETRIS_NO_CHAIN; fi
  timeout -k 10 300 python bench.py --cpu-seconds 0 > $O/bench_$v.json 2> $O/bench_$v.err || { tail $O/bench_$v.err; exit 1; }
  timeout -k 10 300 python bench.py --cpu-seconds 0 --steps 20 --warmup 5 > $O/bench_${v}_20.json 2>/dev/null
  python - <<PY
import json
for f in ("bench_$v","bench_${v}_20"):
    d=json.loads(open(f"$O/{f}.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, "%.2f G/s"%(d["value"]/1e9), "wall %.2f us"%r["launch_us"], "events", r["launch_us_events"], "frac", r["frac"])
PY
done
unset TETRIS_NO_CHAIN
for e in 1 3; do
  TETRIS_LIB=$PWD/drl-tetris_amd/lib/exp_enum$e.so timeout -k 10 200 python profiles/kernel_prof.py enum_noafter > $O/enum_exp$e.json 2> $O/enum_exp$e.err || { tail -5 $O/enum_exp$e.err; exit 1; }
  python -c "import json; d=json.load(open('$O/enum_exp$e.json')); print('enum exp$e noafter', round(d['us_per_launch_events'],2), 'us')"
done
