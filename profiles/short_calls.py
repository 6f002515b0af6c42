#!/usr/bin/env python3
"""Short chained calls as the driver's bench makes them (K launches bracketed by synchronisation), direct dispatch against the stream
path, alternating in one process.  argv: [K, default 20] [players, default 1] [repetitions, default 30]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib

import numpy as np

import __graft_entry__ as ge

ge.package()
mod = importlib.import_module("drl-tetris_amd.distributed")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
sh = mod.ShardedRollout(65536, P, 20, 10, rank=0, world=1, device=0)
for direct in (True, False):
    sh.batch.set_direct_dispatch(direct, min_launches=1)
    sh.run(256, 1)
res = {True: [], False: []}
ev = {True: [], False: []}
for rep in range(reps):
    for direct in (True, False):
        sh.batch.set_direct_dispatch(direct, min_launches=1)
        sh.run(5, 1)
        r = sh.run(K, 1)
        assert sh.batch.rollout_was_direct() == direct
        res[direct].append(r["wall_s"] * 1e6 / K); ev[direct].append(r["event_ms"] * 1e3 / K)
for direct in (True, False):
    w, e = np.array(res[direct]), np.array(ev[direct])
    print(f"K={K} P={P} {'direct ' if direct else 'streams'}: wall per launch median {np.median(w):.3f} min {w.min():.3f} max {w.max():.3f} us;  events median {np.median(e):.3f} us")
sh.close()
