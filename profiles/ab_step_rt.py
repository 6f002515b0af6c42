"""Diagnostic same-box A/B of tetris_step_rt_dev (device arrays in and out) for two builds: profiles/ab_step_rt.py libA.so libB.so"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge
pkg = ge.package()
ptr = lambda t: C.c_void_p(t.data_ptr())
n, K = 65536, 32
gen = torch.Generator(device="cuda").manual_seed(1)
rots = torch.randint(0, 4, (K, n), generator=gen, device="cuda", dtype=torch.uint8)
trans = torch.randint(0, 10, (K, n), generator=gen, device="cuda", dtype=torch.uint8)
for rep in range(3):
    for lib in sys.argv[1:]:
        res = []
        for P in (1, 2):
            who = torch.randint(0, P, (K, n), generator=gen, device="cuda", dtype=torch.uint8)
            b = pkg.TetrisBatch(n, P, 20, 10, seeds=np.arange(n), lib_path=os.path.abspath(lib))
            done = torch.zeros(n, dtype=torch.uint8, device="cuda"); lines = torch.zeros(P * n, dtype=torch.uint8, device="cuda"); dead = torch.zeros(P * n, dtype=torch.uint8, device="cuda")
            t = {4: 0.0, 12: 0.0}
            for r in range(40):
                for w in (4, 12):
                    b.reset(None, seeds=((12345 + 7919 * np.arange(n) + 104729 * r) & 0xFFFF).astype(np.uint16).view(np.int16))
                    b.timer_start()
                    for k in range(w):
                        j = (r * 12 + k) % K
                        b._check(b.lib.tetris_step_rt_dev(b._h, ptr(rots[j]), ptr(trans[j]), ptr(who[j]), 400, ptr(done), ptr(lines), ptr(dead)))
                    t[w] += b.timer_stop() * 1e3
            res.append(f"P={P}: {(t[12] - t[4]) / 40 / 8:.2f} us/call")
            b.close()
        print(os.path.basename(lib), " | ".join(res), flush=True)
