#!/usr/bin/env python3
"""Per-stage device times of split mode (BASELINE config 5) at its per-GPU size: 65 536 two-player games, side 0 and side 1
as two batches on THE ONE GPU of the test box, on one stream, each stage reading the other side's exchange words where that
side's kernel wrote them (what an all-gather would deliver).  Steady-state built-in rollout (policy and auto-reset on the
device).  Prints HIP-event time per step; run under `rocprofv3 --kernel-trace --stats` for the per-kernel rows
(k_split<0> = key interpreter + loop 1, k_split<1> = delayCheck, k_split<2> = winner logic / reset)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json

import numpy as np
import torch

import __graft_entry__ as ge
import importlib

pkg = ge.package()
n, steps = 65536, int(sys.argv[1]) if len(sys.argv) > 1 else 512
seeds = importlib.import_module("drl-tetris_amd.distributed").episode_seeds(0, n)
B = [pkg.TetrisBatch(n, 1, 20, 10, seeds=seeds, device=0, split_side=s) for s in (0, 1)]
stream = torch.cuda.current_stream().cuda_stream
for b in B:
    b.set_stream(stream)
z = lambda: torch.zeros(n, dtype=torch.int32, device="cuda")
A, Bw = [z(), z()], [z(), z()]
words = [B[s].split_words(A[s].data_ptr(), A[1 - s].data_ptr(), Bw[0].data_ptr(), Bw[1].data_ptr()) for s in (0, 1)]


def step3(k):
    """three kernels per side and step (stages A, B, C): what a loop with host-side actions runs"""
    for s in (0, 1):
        B[s].split_rollout_stage(0, k, out=A[s].data_ptr())
    B[0].split_rollout_stage(1, k, words=words[0], out=Bw[0].data_ptr())
    B[1].split_rollout_stage(1, k, words=words[1], out=Bw[1].data_ptr())
    for s in (0, 1):
        B[s].split_rollout_stage(2, k, words=words[s])


def step2(k):
    """two kernels per side and step: stage B, then stage C fused with stage A of the next step (the device-driven rollout)"""
    B[0].split_rollout_stage(1, k, words=words[0], out=Bw[0].data_ptr())
    B[1].split_rollout_stage(1, k, words=words[1], out=Bw[1].data_ptr())
    for s in (0, 1):
        B[s].split_rollout_stage(3, k, words=words[s], out=A[s].data_ptr())


def timed(fn, first, count):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(first, first + count):
        fn(k)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / count


for k in range(64):
    step3(k)
us3 = timed(step3, 64, steps)
k0 = 64 + steps
for s in (0, 1):                                  # the fused form: one plain stage A up front ...
    B[s].split_rollout_stage(0, k0, out=A[s].data_ptr())
for k in range(k0, k0 + 64):
    step2(k)
us2 = timed(step2, k0 + 64, steps)
k1 = k0 + 64 + steps
B[0].split_rollout_stage(1, k1, words=words[0], out=Bw[0].data_ptr())      # ... and one plain stage C at the end
B[1].split_rollout_stage(1, k1, words=words[1], out=Bw[1].data_ptr())
for s in (0, 1):
    B[s].split_rollout_stage(2, k1, words=words[s])
torch.cuda.synchronize()
t0, t1 = B[0].rollout_totals(), B[1].rollout_totals()
print(json.dumps({"games_per_side": n, "steps": steps, "us_per_step_both_sides_one_gpu": us2, "kernels_per_step": 4,
                  "us_per_step_both_sides_one_gpu_three_stage_form": us3, "kernels_per_step_three_stage_form": 6,
                  "env_steps_counted": int(t0[0]), "episodes": int(t0[1]), "lines": int(t0[2] + t1[2]), "sent": int(t0[3] + t1[3])}))
