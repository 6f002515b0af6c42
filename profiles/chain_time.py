#!/usr/bin/env python3
"""Times chained single-step rollout launches of 64k single-player boards with the library in TETRIS_LIB (experiment builds
whose results may be invalid: only the period is read off)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as ge

b = ge.package().TetrisBatch(65536, 1, 20, 10, seeds=np.arange(65536), lib_path=os.environ.get("TETRIS_LIB"))
for rep in range(4):
    ms = b.rollout_launch(2048, 1, first_step=rep * 2048)
    print(os.path.basename(os.environ.get("TETRIS_LIB", "default")), f"{ms * 1e3 / 2048:.2f} us per launch (events)")
