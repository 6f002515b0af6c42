#!/usr/bin/env python3
"""Where one iteration of the drop-in worker loop (profiles/dropin_api.py) spends its time at 4096 envs: each call timed alone."""
import importlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "default" else None
env_mod = importlib.import_module("drl-tetris_amd.environment")
dt = importlib.import_module("drl-tetris_amd.data_types")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
clock = iter(range(1, 1 << 30))
env = env_mod.tetris_environment_vector(n, None, settings={"n_players": 2, "game_size": [20, 10], "seed_source": lambda: next(clock)}, _lib_path=lib)
rng = np.random.default_rng(0)
R, T = rng.integers(0, 4, n), rng.integers(0, 10, n)
lists = [dt.action([8] * int(r) + [2] + [3] * int(t) + [7]) for r, t in zip(R, T)]
who = np.zeros(n, np.int64)


def t(f, k=200, fresh=False):
    """microseconds per call; fresh: the games are reset (untimed) every 8 calls, so that the calls step live games"""
    for _ in range(5):
        f()
    total = 0.0
    for i in range(k):
        if fresh and i % 8 == 0:
            env.reset()
        t0 = time.perf_counter()
        f()
        total += time.perf_counter() - t0
    return total / k * 1e6


B = env.backend
keys, lens = env._pack(lists, who, n)
done_list = [False] * n
out = {"n_envs": n, "us": {
    "backend.snapshot (kernel + copy + numpy array)": t(lambda: B.snapshot()),
    "get_state (snapshot + lazy list)": t(lambda: env.get_state()),
    "_pack of a Python action list": t(lambda: env._pack(lists, who, n)),
    "_pack of an action_batch": t(lambda: env._pack(dt.action_batch.from_rt(R, T), who, n)),
    "action_batch.from_rt": t(lambda: dt.action_batch.from_rt(R, T)),
    "backend.step_keys (pre-packed keys)": t(lambda: B.step_keys(keys, lens), fresh=True),
    "backend.step_rt (arrays)": t(lambda: B.step_rt(R.astype(np.uint8), T.astype(np.uint8), 0), fresh=True),
    "perform_action(list)": t(lambda: env.perform_action(lists, player=0), fresh=True),
    "perform_action(action_batch)": t(lambda: env.perform_action(dt.action_batch.from_rt(R, T), player=0), fresh=True),
    "done list -> finished indices (user code)": t(lambda: [i for i, d in enumerate(done_list) if d]),
    "reset of 100 envs": t(lambda: env.reset(env=list(range(100)))),
    "backend.sync alone": t(lambda: B.sync()),
    "backend.observe_packed": t(lambda: B.observe_packed()),
}}
print(json.dumps(out, indent=1))
