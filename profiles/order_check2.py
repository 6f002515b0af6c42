#!/usr/bin/env python3
"""Why does the chained period of order_check.py wander between 2048-launch runs of one process?  Same alternation, but (a) the RNG
tables are extended first (a scratch batch of the same process runs 30 000 steps: the tables are shared), and (b) every run prints the
library's own host-side enqueue cost (TETRIS_TIMING=1, stderr) beside the wall clock, the HIP events and the table size."""
import os
import sys

os.environ["TETRIS_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib

import __graft_entry__ as ge

ge.package()
mod = importlib.import_module("drl-tetris_amd.distributed")
pre = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
if pre:
    scratch = mod.ShardedRollout(65536, 1, 20, 10, rank=0, world=1, device=0)
    for k in range(pre // 1000):
        scratch.batch.rollout_launch(1000, 1, first_step=k * 1000)
    print("tables after the scratch run:", scratch.batch.table_chunks, "chunks", flush=True)
    scratch.close()
sh = mod.ShardedRollout(65536, 1, 20, 10, rank=0, world=1, device=0)
sh.run(64, 1)
for rep in range(4):
    for chained in (True, True, False):
        sh.batch.set_chained(chained)
        sh.run(8, 1)
        r = sh.run(2048, 1)
        print(f"rep {rep} chained={chained}: wall {r['wall_s'] * 1e6 / 2048:.2f} us  events {r['event_ms'] * 1e3 / 2048:.2f} us  chunks {sh.batch.table_chunks}", flush=True)
sh.close()
