#!/usr/bin/env python3
"""GPU-paced period of chained launches: TETRIS_PREQUEUE=1 parks the chain streams behind a ~5 ms blocker kernel, so all 512
launches of a call are queued before the first starts — the host's launch cost (2.5-5 us per launch, varies between processes
and boxes) does not enter.  argv: P [library build, default = in-tree] [games, default 65536].  Prints pre-queued (HIP events) and host-paced periods."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TETRIS_PREQUEUE"] = "1"
import numpy as np

import __graft_entry__ as ge

pkg = ge.package()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib = sys.argv[2] if len(sys.argv) > 2 else "default"
N = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
b = pkg.TetrisBatch(N, P, 20, 10, seeds=np.arange(N), lib_path=None if lib == "default" else os.path.abspath(lib))
b.rollout_random(256, 1)
step = 256
pre, paced = [], []
for rep in range(4):
    _, ms = b.rollout_random(512, 1, first_step=step); step += 512        # pre-queued (<= 600 launches)
    _, ms2 = b.rollout_random(2048, 1, first_step=step); step += 2048     # host-paced
    pre.append(ms * 1e3 / 512); paced.append(ms2 * 1e3 / 2048)
print(f"{os.path.basename(lib):18s} P={P} N={N} pre-queued " + " ".join(f"{x:5.2f}" for x in pre) + "   host-paced " + " ".join(f"{x:5.2f}" for x in paced) + f"   (chained={b.rollout_is_chained(1)})")
b.close()
