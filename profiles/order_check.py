#!/usr/bin/env python3
"""Does the order of measurements inside one process matter?  Alternates chained / un-chained runs of 2048 single-step launches
(64k boards) several times and prints the wall clock per launch of each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib

import __graft_entry__ as ge

ge.package()
mod = importlib.import_module("drl-tetris_amd.distributed")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sh = mod.ShardedRollout(65536, P, 20, 10, rank=0, world=1, device=0)
sh.run(64, 1)
for rep in range(3):
    for chained in (True, True, False, False):
        sh.batch.set_chained(chained)
        sh.run(8, 1)
        r = sh.run(2048, 1)
        print(f"rep {rep} chained={chained}: wall {r['wall_s'] * 1e6 / 2048:.2f} us  events {r['event_ms'] * 1e3 / 2048:.2f} us")
sh.close()
