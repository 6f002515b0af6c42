#!/usr/bin/env python3
"""XCD-affine launches in a process with many queues: 40 small batches (a HIP stream each) keep stepping in between, the big batch makes 300
chained calls of 40 launches.  A queue's start XCD moves when the driver re-maps hardware queues; calls that meet a moved start are
finished un-chained (slow, exact).  Prints how many calls ran affine, how many took more than 3x the median, and compares every board
with the oracle."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as ge
from oracle import oracle as orc
from tests import engines

pkg = ge.package()
n = 65536
seeds = orc.episode_seed(np.arange(n), 0)
eng = pkg.TetrisBatch(n, 1, 20, 10, seeds=seeds, device=0)
ref = orc.OracleBatch(n, 1, 20, 10, seeds=seeds)
small = [pkg.TetrisBatch(64, 1, 20, 10, seeds=np.arange(64), device=0) for _ in range(40)]
rng = np.random.default_rng(1)
total, step, affine, times = np.zeros(4, np.uint64), 0, 0, []
for call in range(300):
    for sb in small[call % 4::4]:
        sb.step_rt(rng.integers(0, 4, 64).astype(np.uint8), rng.integers(0, 10, 64).astype(np.uint8))
    if call % 50 == 25:                                   # churn: some of the small batches are replaced
        for k in range(0, 40, 5):
            small[k].close()
            small[k] = pkg.TetrisBatch(64, 1, 20, 10, seeds=np.arange(64), device=0)
    t0 = time.perf_counter()
    c, _ = eng.rollout_random(40, 1, first_step=step)
    times.append((time.perf_counter() - t0) * 1e6)
    affine += eng.rollout_was_affine()
    total += c
    step += 40
times = np.array(times)
print(f"300 calls of 40 launches: {affine} ran the affine kernel; call time median {np.median(times):.0f} us, max {times.max():.0f} us, "
      f"{int((times > 3 * np.median(times)).sum())} calls above 3x the median; errors {eng.take_errors()}, still chained {eng.rollout_is_chained(1)}", flush=True)
_, want = ref.rollout_random(step, threads=min(32, len(os.sched_getaffinity(0))))
ok = total.tolist() == want.tolist()
for lo in range(0, n, 8192):
    try:
        engines.assert_same_state(eng, ref, idx=np.arange(lo, lo + 8192, dtype=np.int32), where=f"games {lo}..")
    except AssertionError as e:
        ok = False
        print("MISMATCH", str(e)[:200])
        break
print("bit-exact vs oracle:", ok)
