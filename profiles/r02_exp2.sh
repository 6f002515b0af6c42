#!/bin/bash
# same-box experiments: row padding of the state arrays x chained launches; enumerate store / workgroup variants
set -e -o pipefail
O=gpurun_out/${OUT:-r02y}
mkdir -p $O
for pad in 0 64 192 320 1088 4160; do
for v in chain nochain; do
  if [ $v = nochain ]; then export TETRIS_NO_CHAIN=1; else unset TETRIS_NO_CHAIN; fi
  export TETRIS_ROW_PAD=$pad
  timeout -k 10 300 python bench.py --cpu-seconds 0 > $O/bench_pad${pad}_$v.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$O/bench_pad${pad}_$v.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("pad $pad $v", "%.2f G/s"%(d["value"]/1e9), "wall %.2f us"%r["launch_us"], "frac %.3f"%r["frac"])
PY
done; done
unset TETRIS_NO_CHAIN TETRIS_ROW_PAD
for e in "" enum_plain enum_b16 enum_b64; do
  if [ -n "$e" ]; then export TETRIS_LIB=$PWD/drl-tetris_amd/lib/exp_$e.so; else unset TETRIS_LIB; fi
  for c in enum_planar enum_noafter_planar; do
    timeout -k 10 200 python profiles/kernel_prof.py $c > $O/${c}_$e.json 2> $O/enum.err || { tail -5 $O/enum.err; exit 1; }
    python -c "import json; d=json.load(open('$O/${c}_$e.json')); print('$c', '${e:-default}', round(d['us_per_launch_events'],2), 'us', round(d['frac_of_8TBps'],3))"
  done
done
