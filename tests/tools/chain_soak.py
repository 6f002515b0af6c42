#!/usr/bin/env python3
"""argv: [launches [players [library build]]].  Soak of the chained rollout launches: 65 536 boards x N single-step launches (default 10 000 = 6.6e8 env-steps,
~3e7 episodes) on the rotating chain streams with per-wave epoch hand-over, then counters and EVERY board's complete state against
the oracle on all host cores.  A stale read anywhere in the hand-over would show up here as a diverging board."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

import __graft_entry__ as ge
from oracle import oracle as orc
from tests import engines

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = 65536
seeds = orc.episode_seed(np.arange(n), 0)
lib = os.path.abspath(sys.argv[3]) if len(sys.argv) > 3 else None
eng = ge.package().TetrisBatch(n, P, 20, 10, seeds=seeds, device=0, lib_path=lib)
ref = orc.OracleBatch(n, P, 20, 10, seeds=seeds)
assert eng.rollout_is_chained(1)
t0 = time.time()
c, ms = eng.rollout_random(steps, 1)
print(f"gpu: {steps} chained launches, {ms * 1e3 / steps:.2f} us per launch (direct dispatch: {eng.rollout_was_direct()})", flush=True)
_, want = ref.rollout_random(steps, threads=min(32, len(os.sched_getaffinity(0))))
print(f"oracle done after {time.time() - t0:.0f} s", flush=True)
assert c.tolist() == want.tolist(), (c.tolist(), want.tolist())
for lo in range(0, n, 8192):
    engines.assert_same_state(eng, ref, idx=np.arange(lo, lo + 8192, dtype=np.int32), where=f"games {lo}..")
assert eng.take_errors() == 0 and eng.rollout_is_chained(1)          # no capacity error, and no launch ever fell back
print(f"soak ok (P={P}):", {k: int(v) for k, v in zip(("env_steps", "episodes", "lines", "sent"), c)})
