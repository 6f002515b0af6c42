#!/usr/bin/env python3
"""One secondary kernel, launched back to back with pre-built ctypes arguments (so the Python loop costs ~1 us per launch and
does not bound what is measured); prints HIP-event time per launch.  Run it under `rocprofv3 --kernel-trace --stats` for the
kernel's own duration (profiles/r02_kernels.sh):   python profiles/kernel_prof.py <case> [reps]
cases: enum_rows enum_planar enum_noafter enum_noafter_planar observe step_auto_1p step_auto_2p step_obs_1p step_obs_2p loop_1p loop_2p
(step_obs = tetris_step_rt_observe_dev, one launch; loop = tetris_step_rt_dev_ex + tetris_observe_packed_dev, two launches: the same work)"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import __graft_entry__ as ge

pkg = ge.package()
LIB = os.environ.get("TETRIS_LIB")      # experiment builds (profiles/): another build of the same library
case = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dv = dict(dtype=torch.uint8, device="cuda")
ptr = lambda t: C.c_void_p(t.data_ptr())
out = {"case": case, "reps": reps}

if case.startswith("enum"):
    n = 16384
    b = pkg.TetrisBatch(n, 1, 20, 10, seeds=np.arange(n), lib_path=LIB)
    b.rollout_random(12, 1)                                  # boards taken at step 12
    valid, land, cleared = torch.zeros(n * 40, **dv), torch.zeros(n * 40, dtype=torch.int8, device="cuda"), torch.zeros(n * 40, **dv)
    after = torch.zeros(n * 400, dtype=torch.int32, device="cuda")
    fn = b.lib.tetris_enumerate_drops_dev_ex
    args = (b._h, None, n, None, ptr(valid), ptr(land), ptr(cleared), None if case.startswith("enum_noafter") else ptr(after), 1 if case.endswith("planar") else 0)
    nbytes = n * 44 + n * 40 * (3 if case.startswith("enum_noafter") else 43)
    unit = ("afterstates", n * 40)
elif case == "observe":
    n = 65536
    b = pkg.TetrisBatch(n, 2, 20, 10, seeds=np.arange(n), lib_path=LIB)
    b.rollout_random(12, 1)
    visual, vector, piece = torch.zeros(2 * n * 200, **dv), torch.zeros(2 * n * 12, **dv), torch.zeros(2 * n, **dv)
    fn = b.lib.tetris_observe_packed_dev
    args = (b._h, None, n, None, ptr(visual), ptr(vector), ptr(piece))
    nbytes = 2 * n * (44 + 213)
    unit = ("player_boards", 2 * n)
else:
    P = 1 if case.endswith("1p") else 2
    n = 65536
    b = pkg.TetrisBatch(n, P, 20, 10, seeds=np.arange(n), lib_path=LIB)
    gen = torch.Generator(device="cuda").manual_seed(1)
    rot = torch.randint(0, 4, (n,), generator=gen, device="cuda", dtype=torch.uint8)
    trans = torch.randint(0, 10, (n,), generator=gen, device="cuda", dtype=torch.uint8)
    who = torch.randint(0, P, (n,), generator=gen, device="cuda", dtype=torch.uint8)
    done, lines, dead = torch.zeros(n, **dv), torch.zeros(P * n, **dv), torch.zeros(P * n, **dv)
    fn = b.lib.tetris_step_rt_dev_ex
    args = (b._h, ptr(rot), ptr(trans), ptr(who), 400, ptr(done), ptr(lines), ptr(dead), 1)       # TETRIS_STEP_AUTO_RESET
    nbytes = (389 if P == 1 else 774) * n
    if case.startswith("step_obs") or case.startswith("loop"):
        visual, vector, piece = torch.zeros(P * n * 200, **dv), torch.zeros(P * n * 12, **dv), torch.zeros(P * n, **dv)
        nxt = (1 - who) if P == 2 else who
        nbytes += P * n * 213                                 # + the packed observation written (the step's state is not read again)
        if case.startswith("step_obs"):
            fn = b.lib.tetris_step_rt_observe_dev
            args = args + (ptr(nxt), ptr(visual), ptr(vector), ptr(piece))
        else:
            step_fn, step_args, obs_args = fn, args, (b._h, None, n, ptr(nxt), ptr(visual), ptr(vector), ptr(piece))
            obs_fn = b.lib.tetris_observe_packed_dev
            def fn(*_):
                rc = step_fn(*step_args)
                return rc or obs_fn(*obs_args)
    unit = ("env_steps", n)
    reps = max(reps, 2048)                                   # steady state: resets of finished games inside the launches
torch.cuda.synchronize()
for _ in range(64):
    assert fn(*args) == 0
b.sync()
b.timer_start()
for _ in range(reps):
    fn(*args)
us = b.timer_stop() * 1e3 / reps
b.sync()
out.update({"us_per_launch_events": us, unit[0] + "_per_s": unit[1] / (us * 1e-6), "algorithmic_bytes": nbytes, "GBps": nbytes / (us * 1e-6) / 1e9,
            "frac_of_8TBps": nbytes / (us * 1e-6) / 8e12})
print(json.dumps(out))
b.close()
