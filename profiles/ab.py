"""Diagnostic A/B of two builds of the library on the same GPU box: profiles/ab.py libA.so libB.so"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.package()
libs = sys.argv[1:]
for rep in range(3):
    for lib in libs:
        out = []
        for P in (1, 2):
            b = pkg.TetrisBatch(65536, P, 20, 10, seeds=np.arange(65536), lib_path=lib)
            b.rollout_random(64, 1)
            _, ms1 = b.rollout_random(2048, 1, first_step=64)
            _, ms32 = b.rollout_random(128, 32, first_step=5000)
            out.append(f"P={P}: {ms1 * 1e3 / 2048:.2f} us/launch, fused {ms32 * 1e3 / 128 / 32:.2f} us/step")
            b.close()
        print(os.path.basename(lib), " | ".join(out))
