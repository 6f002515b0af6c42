"""PMC calibration workload: launches of the step kernel with ZERO env-steps (state load + store only), whose byte
count is known exactly: (26 hot words per player + 4 game words) x 4 B x N games in, the same out."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
b = ge.package().TetrisBatch(65536, P, 20, 10, seeds=np.arange(65536))
b.rollout_random(256, 0)
b.close()
