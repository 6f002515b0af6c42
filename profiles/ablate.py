"""Diagnostic: time the rollout kernel built with parts compiled out (-DTE_ABLATE=bits; results of such
builds are wrong by construction, only the time matters)."""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.package()
names = {0: "full", 1: "-philox", 2: "-rot/slide", 4: "-tick", 8: "-settle(clear+spawn)", 16: "-autoreset", 32: "-rot/slide/drop", 63: "-everything"}
for lib in sorted(glob.glob(sys.argv[1] + "/lib_*.so"), key=lambda f: int(f.split("_")[-1][:-3])):
    bits = int(lib.split("_")[-1][:-3])
    b = pkg.TetrisBatch(65536, 1, 20, 10, seeds=np.arange(65536), lib_path=lib)
    try:
        b.rollout_random(64, 1)
        _, ms1 = b.rollout_random(1024, 1, first_step=1000)
        _, ms8 = b.rollout_random(256, 8, first_step=100000)
        print(f"{names.get(bits, bits):24s} 1 step/launch {ms1 * 1e3 / 1024:6.2f} us   per fused step {ms8 * 1e3 / 256 / 8:6.2f} us")
    except Exception as e:
        print(names.get(bits, bits), "error", e)
    b.close()
