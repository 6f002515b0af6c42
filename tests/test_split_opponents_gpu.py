"""Split-mode kernels on a real GPU.  The GPU box has one card, so both sides run in one process (two threads, device
tensors, the library's HIP kernels) with an in-process stand-in for the all-gather; the RCCL path itself is the same
SplitOpponents code with torch.distributed's `nccl` backend (covered functionally by tests/test_split_opponents_gloo.py)."""
import threading

import numpy as np
import pytest

from oracle import oracle as orc
from tests.test_split_opponents_gloo import FIELDS, _actions

pytestmark = pytest.mark.gpu


class ThreadGather:
    """Looks like torch.distributed to SplitOpponents; ranks are threads of this process."""

    def __init__(self, world):
        self.world = world
        self.slots = [None] * world
        self.barrier = threading.Barrier(world)
        self.local = threading.local()

    def bind(self, rank):
        self.local.rank = rank

    def get_rank(self):
        return self.local.rank

    def get_world_size(self):
        return self.world

    def get_backend(self):
        return "nccl"            # device tensors, the batch on torch's stream

    def all_gather_into_tensor(self, out, mine, group=None):
        import torch
        self.slots[self.local.rank] = mine.clone()
        torch.cuda.current_stream().synchronize()
        self.barrier.wait()
        out.copy_(torch.cat([s.view(-1) for s in self.slots]))
        torch.cuda.current_stream().synchronize()
        self.barrier.wait()


@pytest.mark.parametrize("scenario,pieces", [("random", (0, 1, 2, 3, 4, 5, 6)), ("o_only", (6,))])
def test_split_kernels_on_gpu_match_colocated_oracle(scenario, pieces):
    import importlib

    import tests.test_split_opponents_gloo as base
    mod = importlib.import_module("drl-tetris_amd.distributed")
    N, STEPS = base.N, 200
    tg = ThreadGather(2)
    results, errors = {}, []

    def side_main(side):
        try:
            tg.bind(side)
            so = mod.SplitOpponents(N, side=side, peer=1 - side, dist=tg, seeds=orc.episode_seed(np.arange(N), 0), pieces=pieces)
            dones, episode = [], np.zeros(N, np.int64)
            for s in range(STEPS):
                rot, trans, acting = _actions(s, scenario)
                done, _, _ = so.step_rt(rot, trans, acting)
                dones.append(done)
                idx = np.nonzero(done)[0].astype(np.int32)
                if len(idx):
                    episode[idx] += 1
                    so.reset(idx, orc.episode_seed(idx, episode[idx]))
            results[side] = (so.batch.observe(), np.stack(dones))
            so.close()
        except Exception as e:          # let the other thread out of its barrier
            errors.append(e)
            tg.barrier.abort()

    threads = [threading.Thread(target=side_main, args=(s,)) for s in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    ref = orc.OracleBatch(N, 2, 20, 10, pieces=pieces, seeds=orc.episode_seed(np.arange(N), 0))
    episode, want = np.zeros(N, np.int64), []
    for s in range(STEPS):
        rot, trans, acting = _actions(s, scenario)
        d = ref.step_rt(rot, trans, acting)
        want.append(d.copy())
        idx = np.nonzero(d)[0].astype(np.int32)
        if len(idx):
            episode[idx] += 1
            ref.reset(idx, orc.episode_seed(idx, episode[idx]))
    rec, ro, lw = ref.observe()
    for side in (0, 1):
        (got, gro, glw), dones = results[side]
        w = np.stack(want)
        if not np.array_equal(dones, w):
            st, gm = np.argwhere(dones != w)[0]
            raise AssertionError(f"side {side}: done differs first at step {st}, game {gm}: got {dones[st, gm]} want {w[st, gm]}; "
                                 f"{int((dones != w).sum())} differences in total")
        for f in FIELDS:
            assert np.array_equal(got[f][:, 0], rec[f][:, side]), (side, f)
        assert np.array_equal(got["field"][:, 0] > 0, rec["field"][:, side] > 0)
        assert np.array_equal(gro, ro) and np.array_equal(glw, lw)


@pytest.mark.parametrize("N,STEPS", [(4096, 160), (65536, 96)], ids=["4096x160", "c5_per_gpu_size_65536x96"])
def test_split_rollout_kernels_on_gpu_match_oracle_rollout(N, STEPS):
    """Device-driven split rollout (policy + auto-reset inside the split kernels) on the GPU, both sides in one process.
    The second case is BASELINE config 5's per-GPU size: 65 536 games per side (each side = what one GPU of the 8 holds),
    96 steps, counters and EVERY board of both sides against the oracle's co-located two-player rollout on all host cores."""
    import importlib
    import os

    import tests.test_split_opponents_gloo as base
    mod = importlib.import_module("drl-tetris_amd.distributed")
    tg = ThreadGather(2)
    results, errors = {}, []

    def side_main(side):
        try:
            tg.bind(side)
            so = mod.SplitOpponents(N, side=side, peer=1 - side, dist=tg, seeds=orc.episode_seed(np.arange(N), 0))
            so.rollout(STEPS)
            results[side] = (so.batch.observe(), so.batch.rollout_totals())
            so.close()
        except Exception as e:
            errors.append(e)
            tg.barrier.abort()

    threads = [threading.Thread(target=side_main, args=(s,)) for s in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    ref = orc.OracleBatch(N, 2, 20, 10, seeds=orc.episode_seed(np.arange(N), 0))
    _, want = ref.rollout_random(STEPS, threads=min(32, len(os.sched_getaffinity(0))))
    rec, ro, lw = ref.observe()
    t0, t1 = results[0][1], results[1][1]
    assert int(t0[0]) == int(want[0]) == int(t1[0]) == N * STEPS          # env-steps, counted on the device
    assert int(t0[1]) == int(want[1]) == int(t1[1])
    assert int(t0[2]) + int(t1[2]) == int(want[2]) and int(t0[3]) + int(t1[3]) == int(want[3])
    for side in (0, 1):
        got, gro, glw = results[side][0]
        for f in base.FIELDS:
            assert np.array_equal(got[f][:, 0], rec[f][:, side]), (side, f)
        assert np.array_equal(gro, ro) and np.array_equal(glw, lw)
