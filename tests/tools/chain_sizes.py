#!/usr/bin/env python3
"""Chained rollout launches at batch sizes that are not multiples of a wave's 64 games (and small ones): counters and every
board against the oracle.  python tests/tools/chain_sizes.py [P]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import __graft_entry__ as ge
from oracle import oracle as orc
from tests import engines

pkg = ge.package()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for n in (1, 63, 65, 1000, 50001, 65535):
    seeds = orc.episode_seed(np.arange(n), 0)
    eng = pkg.TetrisBatch(n, P, 20, 10, seeds=seeds, device=0)
    ref = orc.OracleBatch(n, P, 20, 10, seeds=seeds)
    chained = eng.rollout_is_chained(1)
    c1, _ = eng.rollout_random(150, 1)
    c2, _ = eng.rollout_random(5, 10, first_step=150)
    _, want = ref.rollout_random(200, threads=min(32, len(os.sched_getaffinity(0))))
    assert (c1 + c2).tolist() == want.tolist(), (n, (c1 + c2).tolist(), want.tolist())
    for lo in range(0, n, 8192):
        engines.assert_same_state(eng, ref, idx=np.arange(lo, min(n, lo + 8192), dtype=np.int32), where=f"n={n} games {lo}..")
    print(f"n={n} P={P} chained={chained}: ok", flush=True)
    eng.close()
