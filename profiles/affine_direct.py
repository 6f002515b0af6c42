#!/usr/bin/env python3
"""Experiment: the XCD-affine chained kernel (-DTE_EXPERIMENT_AFFINE: a block's games follow the workgroup's XCC_ID, plain stores keep the
state in that XCD's L2) launched through the library's own queues WITHOUT cache maintenance between the launches of a queue
(TETRIS_DIRECT_FENCE=none), against the default build under the same conditions.  argv: library [launches per call, default 2000]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as ge
from oracle import oracle as orc
from tests import engines

lib = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1] != "default" else None
K = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
n = 65536
seeds = orc.episode_seed(np.arange(n), 0)
eng = ge.package().TetrisBatch(n, 1, 20, 10, seeds=seeds, device=0, lib_path=lib)
ref = orc.OracleBatch(n, 1, 20, 10, seeds=seeds)
total = np.zeros(4, np.uint64)
step = 0
per = []
for rep in range(5):
    c, ms = eng.rollout_random(K, 1, first_step=step)
    assert eng.rollout_was_direct()
    total += c
    step += K
    per.append(ms * 1e3 / K)
print(f"{os.path.basename(sys.argv[1]) if lib else 'default':22s} fence={os.environ.get('TETRIS_DIRECT_FENCE', 'agent'):6s} us per launch (events): " + " ".join(f"{x:.3f}" for x in per), flush=True)
_, want = ref.rollout_random(step, threads=min(32, len(os.sched_getaffinity(0))))
ok = total.tolist() == want.tolist()
for lo in range(0, n, 8192):
    try:
        engines.assert_same_state(eng, ref, idx=np.arange(lo, lo + 8192, dtype=np.int32), where=f"games {lo}..")
    except AssertionError as e:
        ok = False
        print("MISMATCH", str(e)[:200])
        break
print("bit-exact vs oracle:", ok, " errors:", eng.take_errors(), " still chained:", eng.rollout_is_chained(1))
