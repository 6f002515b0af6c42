#!/usr/bin/env python3
"""The chained period of consecutive 2048-launch runs in ONE process, nothing else in between (no un-chained runs, no mode switches):
does it drift (clocks) or jump (phase / placement)?  argv: [runs [launches [sleep_ms between runs]]]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib

import __graft_entry__ as ge

ge.package()
mod = importlib.import_module("drl-tetris_amd.distributed")
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
sleep_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
sh = mod.ShardedRollout(65536, 1, 20, 10, rank=0, world=1, device=0)
sh.run(64, 1)
out = []
for r in range(runs):
    if sleep_ms:
        time.sleep(sleep_ms * 1e-3)
    res = sh.run(launches, 1)
    out.append((res["event_ms"] * 1e3 / launches, sh.batch.clock_mhz()))
print(f"{runs} runs of {launches} chained launches, sleep {sleep_ms} ms between (us per launch @ shader MHz right after): " + " ".join(f"{x:.2f}@{m:.0f}" for x, m in out))
sh.close()
