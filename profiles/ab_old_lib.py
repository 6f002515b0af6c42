#!/usr/bin/env python3
"""Same-box A/B against an OLDER build of the library (one build per process: two builds cannot share a process, their device
symbols collide).  argv: <lib path | default> <P>.  Un-chained and (where the build chains them) chained single-step rollouts of 64k
games, HIP-event microseconds per launch over 2048 launches, three repetitions each."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

import numpy as np

import __graft_entry__ as ge

pkg = ge.package()
capi = pkg.capi if hasattr(pkg, "capi") else __import__("importlib").import_module("drl-tetris_amd.capi")
lib = None if sys.argv[1] == "default" else os.path.abspath(sys.argv[1])
P = int(sys.argv[2])
if lib:                                  # an older build lacks the newer entry points: bind what it has
    have = C.CDLL(lib)
    for name in list(capi._SIGNATURES):
        if not hasattr(have, name):
            del capi._SIGNATURES[name]
b = pkg.TetrisBatch(65536, P, 20, 10, seeds=np.arange(65536), lib_path=lib)
b.rollout_random(256, 1)
out = []
step = 256
for chained in (False, True):
    b.set_chained(chained)
    if chained and not b.rollout_is_chained(1):
        out.append("chained: n/a")
        continue
    us = []
    for rep in range(3):
        _, ms = b.rollout_random(2048, 1, first_step=step)
        step += 2048
        us.append(ms * 1e3 / 2048)
    out.append(("chained " if chained else "un-chained ") + " ".join(f"{x:5.2f}" for x in us))
print(f"{os.path.basename(sys.argv[1]):22s} P={P}  " + "   ".join(out))
b.close()
