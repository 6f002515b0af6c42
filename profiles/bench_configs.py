#!/usr/bin/env python3
"""Secondary measurements for the BASELINE.json configs that are not bench.py's headline line:
C3 (64k two-player boards), C4 (drop-afterstate enumeration on 16k boards), the observation kernel and
get_actions.  Device-side times from HIP events around repeated launches; prints one JSON object."""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import __graft_entry__ as ge

pkg = ge.package()
out = {}


def advance(b, steps):
    b.rollout_random(steps, 1)


# C2 / C3: rollout, one env-step per launch
for P in (1, 2):
    b = pkg.TetrisBatch(65536, P, 20, 10, seeds=np.arange(65536))
    advance(b, 64)
    c, ms = b.rollout_random(2048, 1, first_step=64)
    out[f"C{1 + P}_rollout_{P}p_64k"] = {"env_steps_per_s": 65536 * 2048 / (ms * 1e-3), "us_per_launch": ms * 1e3 / 2048,
                                        "player_board_steps_per_s": P * 65536 * 2048 / (ms * 1e-3)}
    b.close()

# C4: enumerate_drops on 16 384 boards taken at step 12 of each episode
b = pkg.TetrisBatch(16384, 1, 20, 10, seeds=np.arange(16384))
advance(b, 12)
valid, land, cleared, after = b.enumerate_drops()
t0 = time.perf_counter()
reps = 20
for _ in range(reps):
    b.enumerate_drops(columns=True)
host_s = (time.perf_counter() - t0) / reps
out["C4_enumerate_drops_16k"] = {"afterstates_per_call": int(16384 * 40), "valid_fraction": float(valid.mean()),
                                 "host_call_s_incl_pcie": host_s, "afterstates_per_s_incl_pcie": 16384 * 40 / host_s}
b.close()

# observation kernel + get_actions, host-inclusive (PCIe + numpy) timings
b = pkg.TetrisBatch(65536, 2, 20, 10, seeds=np.arange(65536))
advance(b, 12)
t0 = time.perf_counter()
for _ in range(5):
    b.observe_packed(player=0)
out["observe_packed_64k_2p"] = {"host_call_s_incl_pcie": (time.perf_counter() - t0) / 5, "bytes_out": 2 * 65536 * (200 + 12 + 1)}
sub = np.arange(4096, dtype=np.int32)
t0 = time.perf_counter()
lists = b.get_actions(sub, player=0)
dt = time.perf_counter() - t0
out["get_actions_4096_boards"] = {"host_call_s_incl_pcie_and_python_lists": dt, "mean_lists_per_board": float(np.mean([len(l) for l in lists])),
                                  "max_lists": int(max(len(l) for l in lists)), "max_keys": int(max(len(a) for l in lists for a in l))}
b.close()
# device-side kernel times (HIP events on the batch's stream, outputs stay in HBM): enumerate_drops and observe_packed
import torch  # noqa: E402

def ptr(t):
    return C.c_void_p(t.data_ptr())

b = pkg.TetrisBatch(16384, 1, 20, 10, seeds=np.arange(16384))
advance(b, 12)
n = 16384
dv = dict(dtype=torch.uint8, device="cuda")
valid, land, cleared = torch.zeros(n * 40, **dv), torch.zeros(n * 40, dtype=torch.int8, device="cuda"), torch.zeros(n * 40, **dv)
after = torch.zeros(n * 400, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
bytes_per_call = n * 44 + n * 40 * 43          # SURVEY: 44 B per board in, 43 B per placement out (valid, land_y, cleared, 10 columns)
for name, with_after, planar in (("rows", True, False), ("planar", True, True), ("no_after", False, False)):
    a_ptr = ptr(after) if with_after else None
    for _ in range(5):
        b.enumerate_drops_dev(n, ptr(valid), ptr(land), ptr(cleared), a_ptr, planar=planar)
    b.timer_start()
    reps = 200
    for _ in range(reps):
        b.enumerate_drops_dev(n, ptr(valid), ptr(land), ptr(cleared), a_ptr, planar=planar)
    us = b.timer_stop() * 1e3 / reps
    nbytes = bytes_per_call if with_after else n * 44 + n * 40 * 3
    out["C4_enumerate_drops_16k_device" + ("" if name == "rows" else "_" + name)] = {
        "us_per_call": us, "afterstates_per_s": n * 40 / (us * 1e-6), "algorithmic_bytes": nbytes,
        "GBps": nbytes / (us * 1e-6) / 1e9, "frac_of_8TBps": nbytes / (us * 1e-6) / 8e12}
b.close()

# the same kernel at four times C4's size: the launch + first-load prologue is amortised over 115 MB of output
b = pkg.TetrisBatch(65536, 1, 20, 10, seeds=np.arange(65536))
advance(b, 12)
n = 65536
valid, land, cleared = torch.zeros(n * 40, **dv), torch.zeros(n * 40, dtype=torch.int8, device="cuda"), torch.zeros(n * 40, **dv)
after = torch.zeros(n * 400, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for _ in range(5):
    b.enumerate_drops_dev(n, ptr(valid), ptr(land), ptr(cleared), ptr(after), planar=True)
b.timer_start()
for _ in range(reps):
    b.enumerate_drops_dev(n, ptr(valid), ptr(land), ptr(cleared), ptr(after), planar=True)
us = b.timer_stop() * 1e3 / reps
nbytes = n * 44 + n * 40 * 43
out["C4x4_enumerate_drops_64k_device_planar"] = {"us_per_call": us, "afterstates_per_s": n * 40 / (us * 1e-6), "algorithmic_bytes": nbytes,
                                                 "GBps": nbytes / (us * 1e-6) / 1e9, "frac_of_8TBps": nbytes / (us * 1e-6) / 8e12}
b.close()
del valid, land, cleared, after

b = pkg.TetrisBatch(65536, 2, 20, 10, seeds=np.arange(65536))
advance(b, 12)
n = 65536
visual, vector, piece = torch.zeros(2 * n * 200, **dv), torch.zeros(2 * n * 12, **dv), torch.zeros(2 * n, **dv)
torch.cuda.synchronize()
for _ in range(5):
    b._check(b.lib.tetris_observe_packed_dev(b._h, None, n, None, ptr(visual), ptr(vector), ptr(piece)))
b.timer_start()
for _ in range(reps):
    b._check(b.lib.tetris_observe_packed_dev(b._h, None, n, None, ptr(visual), ptr(vector), ptr(piece)))
us = b.timer_stop() * 1e3 / reps
bytes_per_call = 2 * n * (44 + 213)
out["observe_packed_64k_2p_device"] = {"us_per_call": us, "player_boards_per_s": 2 * n / (us * 1e-6), "algorithmic_bytes": bytes_per_call,
                                       "GBps": bytes_per_call / (us * 1e-6) / 1e9, "frac_of_8TBps": bytes_per_call / (us * 1e-6) / 8e12}
b.close()

# the NN-policy shape: actions arrive as device arrays, outputs stay on the device (tetris_step_rt_dev_ex with device-side
# auto-reset): STEADY STATE — 64 warm-up steps, then 2048 timed steps with the resets of finished games inside the launches
# (about one game in twenty ends per step), no host synchronisation anywhere in the loop
for P in (1, 2):
    b = pkg.TetrisBatch(65536, P, 20, 10, seeds=np.arange(65536))
    n = 65536
    gen = torch.Generator(device="cuda").manual_seed(1)
    K = 32
    rots = torch.randint(0, 4, (K, n), generator=gen, device="cuda", dtype=torch.uint8)
    trans = torch.randint(0, 10, (K, n), generator=gen, device="cuda", dtype=torch.uint8)
    who = torch.randint(0, P, (K, n), generator=gen, device="cuda", dtype=torch.uint8)
    done, lines, dead = torch.zeros(n, **dv), torch.zeros(P * n, **dv), torch.zeros(P * n, **dv)
    finished = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    for k in range(64):
        b.step_rt_dev(ptr(rots[k % K]), ptr(trans[k % K]), ptr(who[k % K]), ptr(done), ptr(lines), ptr(dead), auto_reset=True)
    b.sync()
    steps = 2048
    b.timer_start()
    for k in range(steps):
        b.step_rt_dev(ptr(rots[k % K]), ptr(trans[k % K]), ptr(who[k % K]), ptr(done), ptr(lines), ptr(dead), auto_reset=True)
    us = b.timer_stop() * 1e3 / steps
    b.sync()
    algo = (389 if P == 1 else 774) * n
    out[f"step_rt_dev_auto_reset_{P}p_64k_steady_state"] = {
        "us_per_call": us, "env_steps_per_s": n / (us * 1e-6), "algorithmic_bytes": algo, "GBps": algo / (us * 1e-6) / 1e9,
        "frac_of_8TBps": algo / (us * 1e-6) / 8e12, "steps_timed": steps, "games_finished_in_last_step": int(done.sum().item())}
    b.close()
print(json.dumps(out))
