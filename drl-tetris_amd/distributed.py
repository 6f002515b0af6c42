"""Multi-GPU driver of the environment: one process per GPU, games sharded by global game id.

Games never interact across games (the reference's only cross-board message, PythonHandle::distributeLines,
PythonHandle.cpp:124-136, stays between the players of ONE game, i.e. inside one lane), so sharding by game needs
no data-path collective at all: every rank steps its own `tetris_batch`; `torch.distributed` (RCCL on GPUs, gloo in
the CPU tests) carries only the barrier, the max-over-ranks timing and the sum of the rollout counters.
The reference's own scale-out is the same shape: N independent worker containers (docker-compose.yaml:27).
"""
import os
import time

import numpy as np

from .capi import TetrisBatch


def episode_seeds(first_game, n_games, episode=0):
    """seed16 = (12345 + 7919 i + 104729 e) mod 65536 (as int16), i = GLOBAL game id."""
    i = np.arange(n_games, dtype=np.int64) + int(first_game)
    return ((12345 + 7919 * i + 104729 * int(episode)) & 0xFFFF).astype(np.uint16).view(np.int16)


class ShardedRollout:
    """Rank-local shard [rank * games_per_rank, (rank + 1) * games_per_rank) of one big synthetic rollout."""

    def __init__(self, games_per_rank, n_players=1, height=20, width=10, rank=None, world=None, device=None, dist=None,
                 lib_path=None):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        self.dist = dist                       # an initialised torch.distributed module, or None for a single process
        self.first_game = self.rank * games_per_rank
        dev = int(os.environ.get("LOCAL_RANK", "0")) if device is None else device
        self.batch = TetrisBatch(games_per_rank, n_players, height, width, seeds=episode_seeds(self.first_game, games_per_rank),
                                 device=dev, lib_path=lib_path)
        self.batch.set_game_offset(self.first_game)
        self.next_step = 0

    def _sync(self):
        if self.dist is not None:
            import torch
            self.dist.barrier()
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        self.batch.sync()

    def run(self, launches, steps_per_launch=1, timed=True):
        """Advances every game of every rank by launches * steps_per_launch env-steps.
        -> dict(counters = sum over ranks [env_steps, episodes, lines, sent], wall_s and event_ms = max over ranks)."""
        self._sync()
        t0 = time.perf_counter()
        counters, ev_ms = self.batch.rollout_random(launches, steps_per_launch, first_step=self.next_step)
        self._sync()
        wall = time.perf_counter() - t0
        self.next_step += launches * steps_per_launch
        counters = counters.astype(np.int64)
        if self.dist is not None:
            import torch
            dev = "cuda" if torch.cuda.is_available() and self.dist.get_backend() == "nccl" else "cpu"
            t = torch.tensor([wall, ev_ms], dtype=torch.float64, device=dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            wall, ev_ms = float(t[0]), float(t[1])
            c = torch.tensor(counters.tolist(), dtype=torch.int64, device=dev)
            self.dist.all_reduce(c, op=self.dist.ReduceOp.SUM)
            counters = c.cpu().numpy()
        return {"counters": counters, "wall_s": wall, "event_ms": ev_ms}

    def close(self):
        self.batch.close()
