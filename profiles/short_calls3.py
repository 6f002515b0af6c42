#!/usr/bin/env python3
"""K-launch calls, alternating in one process: stream path, direct dispatch, direct dispatch with high-priority queues
(three batches of the same games; argv[4] = 1: created in the opposite order — the batch created last tends to be the fastest).  argv: [K] [players] [repetitions]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib

import numpy as np

import __graft_entry__ as ge

ge.package()
mod = importlib.import_module("drl-tetris_amd.distributed")
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
flip = len(sys.argv) > 4 and sys.argv[4] == "1"
cfg = {}
order = (("streams", {}), ("direct", {}), ("direct-high-prio", {"TETRIS_DIRECT_PRIO": "high"}), ("direct-agent-edges", {"TETRIS_DIRECT_EDGE": "agent"}))
for name, env in (order[::-1] if flip else order):
    if name == "direct-agent-edges":
        continue                                         # (the edge knob is read once per process: run separately)
    os.environ.pop("TETRIS_DIRECT_PRIO", None)
    os.environ.update(env)
    sh = mod.ShardedRollout(65536, P, 20, 10, rank=0, world=1, device=0)
    sh.batch.set_direct_dispatch(name != "streams", min_launches=1)
    sh.run(256, 1)                                       # (the queues are made here, with the priority of the moment)
    cfg[name] = sh
res = {k: [] for k in cfg}
for rep in range(reps):
    for name, sh in cfg.items():
        sh.run(5, 1)
        r = sh.run(K, 1)
        assert sh.batch.rollout_was_direct() == (name != "streams")
        res[name].append(r["wall_s"] * 1e6 / K)
for name, w in res.items():
    w = np.array(w)
    print(f"K={K} P={P} {name:12s}: wall per launch median {np.median(w):.3f} min {w.min():.3f} max {w.max():.3f} us")
for sh in cfg.values():
    sh.close()
