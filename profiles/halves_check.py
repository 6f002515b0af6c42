#!/usr/bin/env python3
"""Experiment: 64k games as TWO independent half batches (32k games each), each on its own stream, launched concurrently from two
host threads, against one 64k batch.  Prints the per-env-step period (all 64k games advanced by one step)."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as ge

pkg = ge.package()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = 2048
whole = pkg.TetrisBatch(65536, P, 20, 10, seeds=np.arange(65536))
whole.rollout_random(256, 1)
for rep in range(3):
    _, ms = whole.rollout_random(K, 1, first_step=1000 + rep * K)
    print(f"one batch of 64k games (chained={whole.rollout_is_chained(1)}): {ms * 1e3 / K:.2f} us per env-step")
whole.close()
for chain in (False, True):
    halves = [pkg.TetrisBatch(32768, P, 20, 10, seeds=np.arange(32768) + 32768 * h) for h in range(2)]
    for h in halves:
        h.set_chained(chain)
        h.rollout_random(256, 1)
    for rep in range(3):
        bar = threading.Barrier(3)
        def work(h):
            bar.wait()
            h.rollout_launch(K, 1, first_step=1000 + rep * K)
            bar.wait()
        ts = [threading.Thread(target=work, args=(h,)) for h in halves]
        for t in ts: t.start()
        bar.wait(); t0 = time.perf_counter(); bar.wait(); dt = time.perf_counter() - t0
        for t in ts: t.join()
        print(f"two half batches on two streams, two host threads (set_chained={chain}; chained={[h.rollout_is_chained(1) for h in halves]}): {dt * 1e6 / K:.2f} us per env-step")
    for h in halves: h.close()
