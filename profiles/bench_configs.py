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
for _ in range(5):
    b._check(b.lib.tetris_enumerate_drops_dev(b._h, None, n, None, ptr(valid), ptr(land), ptr(cleared), ptr(after)))
b.timer_start()
reps = 200
for _ in range(reps):
    b._check(b.lib.tetris_enumerate_drops_dev(b._h, None, n, None, ptr(valid), ptr(land), ptr(cleared), ptr(after)))
us = b.timer_stop() * 1e3 / reps
bytes_per_call = n * 44 + n * 40 * 43
out["C4_enumerate_drops_16k_device"] = {"us_per_call": us, "afterstates_per_s": n * 40 / (us * 1e-6), "algorithmic_bytes": bytes_per_call,
                                        "GBps": bytes_per_call / (us * 1e-6) / 1e9, "frac_of_8TBps": bytes_per_call / (us * 1e-6) / 8e12}
b.close()

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

# the NN-policy shape: actions arrive as device arrays, outputs stay on the device (tetris_step_rt_dev), no resets inside the
# timed region (boards that end stay round_over, as between perform_action and reset in the reference's loop)
for P in (1, 2):
    b = pkg.TetrisBatch(65536, P, 20, 10, seeds=np.arange(65536))
    n = 65536
    gen = torch.Generator(device="cuda").manual_seed(1)
    K = 32
    rots = torch.randint(0, 4, (K, n), generator=gen, device="cuda", dtype=torch.uint8)
    trans = torch.randint(0, 10, (K, n), generator=gen, device="cuda", dtype=torch.uint8)
    who = torch.randint(0, P, (K, n), generator=gen, device="cuda", dtype=torch.uint8)
    done, lines, dead = torch.zeros(n, **dv), torch.zeros(P * n, **dv), torch.zeros(P * n, **dv)
    torch.cuda.synchronize()
    # windows of 4 and of 12 launches on freshly reset boards; the slope (t12 - t4) / 8 is the cost of one launch without the
    # event pair and the pipeline fill of a short window (steps 4..11 of an episode: hardly any game has ended yet)
    t = {4: 0.0, 12: 0.0}
    reps = 40
    for rep in range(reps):
        for w in (4, 12):
            b.reset(None, seeds=((12345 + 7919 * np.arange(n) + 104729 * rep) & 0xFFFF).astype(np.uint16).view(np.int16))
            b.timer_start()
            for k in range(w):
                j = (rep * 12 + k) % K
                b._check(b.lib.tetris_step_rt_dev(b._h, ptr(rots[j]), ptr(trans[j]), ptr(who[j]), 400, ptr(done), ptr(lines), ptr(dead)))
            t[w] += b.timer_stop() * 1e3
    us = (t[12] - t[4]) / reps / 8
    algo = (389 if P == 1 else 774) * n
    out[f"step_rt_dev_{P}p_64k_device"] = {"us_per_call": us, "env_steps_per_s": n / (us * 1e-6), "algorithmic_bytes": algo,
                                            "GBps": algo / (us * 1e-6) / 1e9, "frac_of_8TBps": algo / (us * 1e-6) / 8e12}
    b.close()
print(json.dumps(out))
