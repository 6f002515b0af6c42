#!/usr/bin/env python3
"""Runs 256 un-chained single-step rollout launches of 64k single-player boards with the library given as argv[1] (an ablation
build, -DTE_ABLATE=bits: results of such builds are wrong by construction, only instruction counts and times matter)."""
import os
import sys

os.environ["TETRIS_NO_CHAIN"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as ge

b = ge.package().TetrisBatch(65536, 1, 20, 10, seeds=np.arange(65536), lib_path=sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "full" else None)
b.rollout_random(64, 1)
_, ms = b.rollout_random(256, 1, first_step=1000)
print(f"{ms * 1e3 / 256:.2f} us per launch")
b.close()
