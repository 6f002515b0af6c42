#!/usr/bin/env python3
"""Per-kernel, per-counter statistics of rocprofv3 --pmc counter_collection CSVs -> <dir>/summary.json (+ printed).

Every dispatch is kept as a sample: n, mean, min, median, max per (kernel, counter).  Round 2 kept the mean alone, over 2 128
dispatches of one process (device pre-conditioning, warm-up, timed launches and pre-queued groups behind a blocker kernel), and
that mean was wrong (WRITE_SIZE 4 516 KiB for a kernel that provably stores 7 168 KiB) without anything in the file showing it.
Round 3's runs hold the warm-up and the timed launches only (bench.py --precondition-ms 0 --no-gpu-paced), the spread is printed,
and the traffic figure is cross-checked three ways: the chained kernel on one stream (TETRIS_CHAIN_DEPTH=1), on three streams,
and the un-chained kernel — rocprofv3 serialises the dispatches of a --pmc run, and the three agree (profiles/r03/).
Raw counters only; byte figures are derived in make_traffic_json.py."""
import csv
import glob
import json
import os
import statistics
import sys
from collections import defaultdict

d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, cs in acc.items():
    out[k] = {"_dispatches": max(len(v) for v in cs.values())}
    for c, v in cs.items():
        out[k][c] = statistics.fmean(v)                      # (the key a round-2 reader expects: the mean)
        out[k][c + "__stats"] = {"n": len(v), "mean": statistics.fmean(v), "min": min(v), "median": statistics.median(v), "max": max(v)}
json.dump(out, open(os.path.join(d, "summary.json"), "w"), indent=1)
for k, cs in out.items():
    if any(t in k for t in ("k_game", "k_chain", "k_duo", "k_split")):
        print(k)
        for c, v in sorted(cs.items()):
            if c.endswith("__stats"):
                print(f"   {c[:-7]:28s} n {v['n']:5d}  mean {v['mean']:14.1f}  min {v['min']:14.1f}  median {v['median']:14.1f}  max {v['max']:14.1f}")
