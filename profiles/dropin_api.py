#!/usr/bin/env python3
"""Throughput of the drop-in Python API in the shape of the reference's worker loop (drl_tetris/worker.py:91-118):

    state = env.get_state();  reward, done = env.perform_action(action, player=p);  s_prime = env.get_state()
    experience = (state, action, reward, s_prime, p, done)  -> buffer;   env.reset(env=[i for i, d in enumerate(done) if d])

through drl-tetris_amd/environment.py (tetris_environment_vector), two players, 20x10, random (rotation, translation) actions.
Variants:  "lists"    actions are a Python list of `action` objects (the reference's own calling convention);
           "batch"    actions are one data_types.action_batch (the same keys as arrays: what an agent that decides for all envs at
                      once hands over);
           "inspect"  as "lists", and every state of every iteration is looked at the way an agent does (state[player] -> the
                      state_dict of state_processors.py:23-54): the per-env Python path end to end.
argv: [lib path | "default"] -> one JSON line."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib

import numpy as np

lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "default" else None
env_mod = importlib.import_module("drl-tetris_amd.environment")
dt = importlib.import_module("drl-tetris_amd.data_types")


def run(n, variant, seconds):
    clock = iter(range(1, 1 << 30))
    env = env_mod.tetris_environment_vector(n, None, settings={"n_players": 2, "game_size": [20, 10], "seed_source": lambda: next(clock)}, _lib_path=lib)
    rng = np.random.default_rng(0)
    buffer, p, steps, episodes = [], 0, 0, 0
    # the actions of 64 iterations are drawn up front: the policy is not what is being timed
    R, T = rng.integers(0, 4, (64, n)), rng.integers(0, 10, (64, n))
    acts = [[dt.action([8] * int(r) + [2] + [3] * int(t) + [7]) for r, t in zip(R[k], T[k])] for k in range(64)] if variant != "batch" else \
           [dt.action_batch.from_rt(R[k], T[k]) for k in range(64)]
    for warm in (True, False):
        t0, it = time.perf_counter(), 0
        while time.perf_counter() - t0 < (0.3 if warm else seconds):
            p = 1 - p
            state = env.get_state()
            if variant == "inspect":
                obs = [s[p] for s in state]
            a = acts[it % 64]
            if variant == "batch" and "_make" not in a.__dict__:
                a = acts[it % 64] = dt.action_batch.from_rt(R[it % 64], T[it % 64])
            reward, done = env.perform_action(a, player=p)
            s_prime = env.get_state()
            buffer.append((state, a, reward, s_prime, p, done))
            if len(buffer) > 8:
                buffer.pop(0)
            finished = [i for i, d in enumerate(done) if d]
            env.reset(env=finished)
            it += 1
            if not warm:
                episodes += len(finished)
        if not warm:
            dt_s = time.perf_counter() - t0
            steps = it * n
    env.backend.close()
    return {"n_envs": n, "variant": variant, "env_steps_per_s": steps / dt_s, "iterations": it, "ms_per_iteration": dt_s * 1e3 / it, "episodes": episodes}


out = {"what": "worker-loop shape through tetris_environment_vector (get_state, perform_action, get_state, reset of finished envs), 2 players, 20x10",
       "library": lib or "drl-tetris_amd/lib/libtetris_hip.so", "rows": []}
for n in (32, 4096):
    for variant in ("lists", "batch", "inspect"):
        out["rows"].append(run(n, variant, 2.0 if variant != "inspect" else 1.0))
print(json.dumps(out))
