#!/bin/bash
# secondary kernels: event-timed back-to-back launches, then the same under rocprofv3 --kernel-trace --stats (kernel durations)
set -e -o pipefail
O=gpurun_out/${OUT:-r02k}
mkdir -p $O
R=$GRAFT_REPO_ROOT
for c in ${CASES:-enum_rows enum_planar enum_noafter observe step_auto_1p step_auto_2p}; do
  timeout -k 10 200 python profiles/kernel_prof.py $c > $O/$c.json 2> $O/$c.err || { tail -5 $O/$c.err; exit 1; }
  (cd /tmp; export TMPDIR=/tmp; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$c -- python3 $R/profiles/kernel_prof.py $c > $R/$O/prof_$c.log 2>&1)
  python - <<PY
import json, glob, csv
d=json.load(open("$O/$c.json"))
print("$c", "events %.2f us"%d["us_per_launch_events"], "frac %.3f"%d["frac_of_8TBps"])
for f in glob.glob("$O/prof_$c/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:2]: print("   rocprof", r["Name"][:60], r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"])
PY
done
