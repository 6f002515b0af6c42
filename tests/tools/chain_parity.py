#!/usr/bin/env python3
"""Parity of an experiment build of the library (TETRIS_LIB=path): 65 536 boards (argv[1] players each), 64 chained single-step launches
+ 4 fused ones, counters and every board against the oracle on all host cores."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import __graft_entry__ as ge
from oracle import oracle as orc
from tests import engines

pkg = ge.package()
n = 65536
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
seeds = orc.episode_seed(np.arange(n), 0)
eng = pkg.TetrisBatch(n, P, 20, 10, seeds=seeds, device=0, lib_path=os.environ.get("TETRIS_LIB"))
ref = orc.OracleBatch(n, P, 20, 10, seeds=seeds)
print("chained:", eng.rollout_is_chained(1), eng.rollout_is_chained(8))
c1, _ = eng.rollout_random(64, 1)
c2, _ = eng.rollout_random(4, 8, first_step=64)
_, want = ref.rollout_random(96, threads=min(32, len(os.sched_getaffinity(0))))
assert (c1 + c2).tolist() == want.tolist(), ((c1 + c2).tolist(), want.tolist())
for lo in range(0, n, 8192):
    engines.assert_same_state(eng, ref, idx=np.arange(lo, lo + 8192, dtype=np.int32), where=f"games {lo}..")
print("parity ok:", os.environ.get("TETRIS_LIB", "default library"))
