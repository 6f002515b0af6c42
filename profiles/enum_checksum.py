#!/usr/bin/env python3
"""Checksum of k_enumerate's outputs (16 384 boards at step 12, rows and planar layouts) for the library build in TETRIS_LIB
(default: in-tree): experiment builds must print the same line as the shipped one."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as ge

b = ge.package().TetrisBatch(16384, 1, 20, 10, seeds=np.arange(16384), lib_path=os.environ.get("TETRIS_LIB"))
b.rollout_random(12, 1)
h = hashlib.sha256()
v, y, c, a = b.enumerate_drops()
ok = v.astype(bool)
for arr in (v, np.where(ok, y, 0), np.where(ok, c, 0), np.where(ok[..., None], a, 0)):
    h.update(np.ascontiguousarray(arr).tobytes())
print("enumerate checksum", h.hexdigest()[:16], "valid fraction", float(ok.mean()))
b.close()
