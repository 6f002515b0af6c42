"""Diagnostic: launch floor of the step kernel (state load + store, zero env-steps) vs 1 and 2 steps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.package()
for P in (1, 2):
    b = pkg.TetrisBatch(65536, P, 20, 10, seeds=np.arange(65536))
    b.rollout_random(64, 1)
    for S in (0, 1, 2, 4, 8):
        c, ms = b.rollout_random(1024, S, first_step=10000 * (S + 1))
        print(f"P={P} steps/launch={S}: {ms * 1e3 / 1024:.2f} us per launch")
    b.close()
