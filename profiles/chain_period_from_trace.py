#!/usr/bin/env python3
"""The period of chained launches from rocprofv3's own dispatch timestamps (an independent clock: not HIP events, not the host).

usage: chain_period_from_trace.py <dir with *_kernel_trace.csv of a TETRIS_PREQUEUE=1 run> [out.json]

A pre-queued call parks the chain streams behind k_blocker until the host has queued every launch of the call, so the launches that
follow a blocker in the trace are GPU-paced whatever a launch costs the host under the profiler.  For every such group:
  period_us            = (end of the last launch - start of the first) / launches
  start_to_start_us    = median difference between consecutive starts (all queues merged, in start order)
  per_queue            = launches and median start-to-start per hardware queue (each queue carries every depth-th launch)
  duration_us          = median / mean of the kernel's own duration (longer than the period: launches overlap and a wave's wait
                         for its predecessor lies inside its kernel's duration)."""
import csv
import glob
import json
import os
import statistics
import sys

d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
groups, cur, armed = [], [], False
for st, en, name, q in rows:
    if "k_blocker" in name:
        if cur:
            groups.append(cur)
        cur, armed = [], True
    elif armed and ("k_chain" in name or "k_duo" in name):
        cur.append((st, en, name, q))
    elif cur:
        groups.append(cur)
        cur, armed = [], False
if cur:
    groups.append(cur)
out = {"source": d, "clock": "rocprofv3 --kernel-trace dispatch timestamps (ns)", "groups": []}
for g in groups:
    if len(g) < 32:
        continue
    starts = [x[0] for x in g]
    per_q = {}
    for q in sorted(set(x[3] for x in g)):
        qs = [x[0] for x in g if x[3] == q]
        per_q[str(q)] = {"launches": len(qs), "start_to_start_us_median": statistics.median(b - a for a, b in zip(qs, qs[1:])) / 1e3 if len(qs) > 1 else None}
    dur = [(x[1] - x[0]) / 1e3 for x in g]
    out["groups"].append({
        "kernel": g[0][2], "launches": len(g),
        "period_us": (max(x[1] for x in g) - starts[0]) / len(g) / 1e3,
        "start_to_start_us_median": statistics.median(b - a for a, b in zip(starts, starts[1:])) / 1e3,
        "duration_us_median": statistics.median(dur), "duration_us_mean": statistics.fmean(dur),
        "hardware_queues": per_q,
    })
big = [g for g in out["groups"] if g["launches"] >= 256]
if big:
    out["period_us_of_the_512_launch_groups"] = [round(g["period_us"], 3) for g in big]
    out["period_us_best"] = min(g["period_us"] for g in big)
print(json.dumps(out, indent=1))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
