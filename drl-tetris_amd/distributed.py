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

    def _drain(self):
        """This rank's launches have finished: tetris_rollout_launch returns only after it has seen the end events of every stream
        it used (it polls them: a blocking wait is woken through an interrupt 10-20 us late), and tetris_sync re-checks that the
        batch's streams are drained.  Deliberately NOT torch.cuda.synchronize(): a device-wide synchronise costs ~45 us on
        this runtime (measured with one RCCL rank: 146 against 101 us for the driver's 20-launch region) although nothing but these
        launches is in flight; the device-wide synchronise of the closing bracket follows in _sync(), after the clock has stopped."""
        self.batch.sync()

    def run(self, launches, steps_per_launch=1, timed=True):
        """Advances every game of every rank by launches * steps_per_launch env-steps.
        -> dict(counters = sum over ranks [env_steps, episodes, lines, sent], wall_s and event_ms = max over ranks).
        Only the step-kernel launches lie inside the clocked region (barrier + synchronize on both sides of it; the closing
        barrier itself is not clocked, see below): the
        counters are per-game words kept by the kernels themselves (env-steps are counted on the device) and are summed by a
        separate kernel before and after, outside the region."""
        before = self.batch.rollout_totals()
        self._sync()                                   # opening bracket: barrier over the ranks + synchronize
        t0 = time.perf_counter()
        ev_ms = self.batch.rollout_launch(launches, steps_per_launch, first_step=self.next_step)
        self._drain()                                  # this rank's launches have finished (the call above already waited for them)
        wall = time.perf_counter() - t0
        self._sync()                                   # closing bracket; the clock stops BEFORE it: a barrier over RCCL costs tens of
                                                       # microseconds, as much as a short rollout, and is no part of the K steps — the
                                                       # reported time is the MAX over ranks of each rank's own start-to-drained time
        counters = (self.batch.rollout_totals() - before).astype(np.int64)
        self.next_step += launches * steps_per_launch
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


class SplitOpponents:
    """Cross-device opponents (BASELINE config 5): this rank holds ONE player (`side`) of n_games two-player games,
    rank `peer` holds the other player of the same games.  The garbage-line exchange of the reference
    (PythonHandle::distributeLines, PythonHandle.cpp:124-136) and the dead flags of its winner logic become three
    all-gathers of one 32-bit word per board per step (stage protocol: csrc/tetris_engine.h "split mode").
    The all-gathers run in a process group of the two ranks of the pair only: the words concern nobody else, and over the
    whole world every GPU would receive world - 1 rows to read one (xGMI is point-to-point: at 8 ranks 7 links' worth of
    traffic for 1).  `pairs` = every pair of the job, the same list on every rank (default: ranks 2k and 2k + 1) — creating
    a process group is a collective over the world.

    With the `nccl` backend (= RCCL over xGMI) every buffer is a device tensor and the batch runs on torch's current
    stream, so kernels and collectives are stream-ordered without host synchronisation.  With `gloo` (CPU tests, where
    the library is the CPU test harness) the same code runs on host tensors.
    """

    def __init__(self, n_games, side, peer, dist, height=20, width=10, pieces=(0, 1, 2, 3, 4, 5, 6), seeds=None, device=0,
                 lib_path=None, pairs=None):
        import torch

        self.torch, self.dist = torch, dist
        self.n, self.side, self.peer = int(n_games), int(side), int(peer)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        if pairs is None:
            pairs = [(2 * k, 2 * k + 1) for k in range(self.world // 2)]
        mine = tuple(sorted((self.rank, self.peer)))
        pairs = [tuple(sorted(p)) for p in pairs]
        if mine not in pairs:
            raise ValueError(f"rank {self.rank} and its peer {self.peer} are not one of the job's pairs {pairs}")
        self.group = None
        for p in pairs:                                   # (every rank creates every group, in the same order)
            g = dist.new_group(list(p)) if self.world > 2 else None      # two ranks: the world is the pair
            if p == mine:
                self.group = g
        self.row_mine, self.row_peer = (0, 1) if self.rank < self.peer else (1, 0)     # rows of a gather buffer: group ranks ascending
        self.on_gpu = dist.get_backend() == "nccl"
        self.dev = torch.device("cuda", device) if self.on_gpu else torch.device("cpu")
        self.batch = TetrisBatch(n_games, 1, height, width, pieces=pieces, seeds=seeds, device=device, lib_path=lib_path,
                                 split_side=side)
        if self.on_gpu:
            self.batch.set_stream(torch.cuda.current_stream(self.dev).cuda_stream)
        z = lambda *shape, dt=torch.uint8: torch.zeros(*shape, dtype=dt, device=self.dev)
        self.rot, self.trans, self.acting = z(self.n), z(self.n), z(self.n)
        self.done, self.lines, self.dead = z(self.n), z(self.n), z(self.n)
        # Exchange words: what my kernels write (A, B) and one gather buffer per exchange.  Every stage reads the words where
        # they lie — its own output or the peer's row of a gather buffer — so a step is 3 kernels + 3 all-gathers and nothing
        # else: no copies, no allocation (`zero` is what the side that has nothing to say in an exchange contributes).
        self.a_mine, self.b_mine, self.zero = z(self.n, dt=torch.int32), z(self.n, dt=torch.int32), z(self.n, dt=torch.int32)
        self.g1, self.g2, self.g3 = (z(2, self.n, dt=torch.int32) for _ in range(3))
        B = self.batch
        opp_a = self.g1[self.row_peer].data_ptr()
        b0 = self.b_mine.data_ptr() if self.side == 0 else self.g2[self.row_peer].data_ptr()
        b1 = self.g3[self.row_peer].data_ptr() if self.side == 0 else self.b_mine.data_ptr()
        self.words = B.split_words(self.a_mine.data_ptr(), opp_a, b0, b1)

    def _exchange_and_tick(self, stage1):
        """Exchange 1 (loop-1 words both ways), stage B of player 0, exchange 2 (its tick words), stage B of player 1, exchange 3."""
        gather = lambda out, src: self.dist.all_gather_into_tensor(out, src, group=self.group)
        gather(self.g1.view(-1), self.a_mine)
        if self.side == 0:
            stage1(self.b_mine.data_ptr())
            gather(self.g2.view(-1), self.b_mine)
            gather(self.g3.view(-1), self.zero)
        else:
            gather(self.g2.view(-1), self.zero)
            stage1(self.b_mine.data_ptr())
            gather(self.g3.view(-1), self.b_mine)

    def _step(self, stage0, stage1, stage2):
        """The stage protocol of one step (csrc/tetris_engine.h "split mode"): exchange 1 = loop-1 words both ways,
        exchange 2 = player 0's tick words, exchange 3 = player 1's tick words."""
        stage0(self.a_mine.data_ptr())
        self._exchange_and_tick(stage1)
        stage2()

    def step_rt(self, rot, trans, acting, ms=400):
        """One env-step of all games: player acting[g] of game g plays (rot[g], trans[g]).  -> done, lines, dead (numpy, my side)."""
        t = self.torch
        for dst, src in ((self.rot, rot), (self.trans, trans), (self.acting, acting)):
            dst.copy_(t.as_tensor(np.ascontiguousarray(src, dtype=np.uint8)), non_blocking=True)
        B, w = self.batch, self.words
        self._step(lambda out: B.split_stage(0, rot=self.rot.data_ptr(), trans=self.trans.data_ptr(), acting=self.acting.data_ptr(), out=out, ms=ms),
                   lambda out: B.split_stage(1, words=w, out=out, ms=ms),
                   lambda: B.split_stage(2, words=w, done=self.done.data_ptr(), lines=self.lines.data_ptr(), dead=self.dead.data_ptr(), ms=ms))
        if self.on_gpu:
            t.cuda.current_stream(self.dev).synchronize()
        return self.done.cpu().numpy().copy(), self.lines.cpu().numpy().copy(), self.dead.cpu().numpy().copy()

    def rollout(self, steps, first_step=0, policy_seed=0xD71, ms=400):
        """`steps` env-steps of the built-in synthetic rollout on every game (policy, acting player and auto-reset are
        computed on the device, identically on both sides): per step two kernels and three all-gathers, no copies and no host
        synchronisation in between.  -> seconds (wall, after a final stream sync)."""
        t, B, w = self.torch, self.batch, self.words
        if self.on_gpu:
            t.cuda.current_stream(self.dev).synchronize()
        t0 = time.perf_counter()
        # per step TWO kernels and three all-gathers: stage B, and stage C fused with stage A of the next step (stage 3: one pass
        # over the state instead of two); the call's first step starts with a plain stage A, its last one ends with a plain stage C
        end = first_step + steps
        if steps > 0:
            B.split_rollout_stage(0, first_step, out=self.a_mine.data_ptr(), policy_seed=policy_seed, ms=ms)
        for s in range(first_step, end):
            self._exchange_and_tick(lambda out: B.split_rollout_stage(1, s, words=w, out=out, policy_seed=policy_seed, ms=ms))
            if s + 1 < end:
                B.split_rollout_stage(3, s, words=w, out=self.a_mine.data_ptr(), policy_seed=policy_seed, ms=ms)
            else:
                B.split_rollout_stage(2, s, words=w, policy_seed=policy_seed, ms=ms)
        if self.on_gpu:
            t.cuda.current_stream(self.dev).synchronize()
        return time.perf_counter() - t0

    def reset(self, idx, seeds):
        if self.on_gpu:
            self.torch.cuda.current_stream(self.dev).synchronize()
        self.batch.reset(idx, seeds)

    def close(self):
        self.batch.close()
