"""N>1 path on CPU: two `gloo` ranks, each stepping its shard through the product's ShardedRollout (kernel bodies
via the CPU test harness), must together equal ONE oracle rollout over the union of the games — counters and the
final state of every game — i.e. sharding by global game id is exact and needs no data-path collective."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as ge
from oracle import oracle as orc

N_PER_RANK, P, STEPS = 96, 2, 80


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    sys.path.insert(0, ge.ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mod = __import__("importlib").import_module("drl-tetris_amd.distributed")
    sh = mod.ShardedRollout(N_PER_RANK, P, 20, 10, rank=rank, world=world, device=0, dist=dist, lib_path=ge.build_harness())
    res1 = sh.run(STEPS // 2, 1)
    res2 = sh.run(STEPS // 4, 2)            # continues at the right global step, fused 2 steps per launch
    rec, ro, lw = sh.batch.observe()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rec=rec, ro=ro, lw=lw, c=res1["counters"] + res2["counters"])
    sh.close()
    dist.destroy_process_group()


def test_two_rank_shards_equal_one_oracle_rollout(tmp_path):
    ge.build_harness()
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    n = world * N_PER_RANK
    ref = orc.OracleBatch(n, P, 20, 10, seeds=orc.episode_seed(np.arange(n), 0))
    _, want = ref.rollout_random(STEPS)
    rec, ro, lw = ref.observe()
    for rank in range(world):
        got = np.load(os.path.join(str(tmp_path), f"rank{rank}.npz"))
        assert got["c"].tolist() == [int(x) for x in want]            # every rank holds the all-reduced totals
        sl = slice(rank * N_PER_RANK, (rank + 1) * N_PER_RANK)
        for f in ("x", "y", "next", "dead", "time_ms", "piece_draws", "lines_sent", "lines_cleared", "incoming"):
            assert np.array_equal(got["rec"][f], rec[f][sl]), (rank, f)
        assert np.array_equal(got["rec"]["field"] > 0, rec["field"][sl] > 0)
        assert np.array_equal(got["ro"], ro[sl]) and np.array_equal(got["lw"], lw[sl])
    # and a shard stepped alone (no process group) gives the same per-game results as inside the group
    mod = __import__("importlib").import_module("drl-tetris_amd.distributed")
    solo = mod.ShardedRollout(N_PER_RANK, P, 20, 10, rank=1, world=2, device=0, dist=None, lib_path=ge.build_harness())
    solo.run(STEPS, 1)
    assert np.array_equal(solo.batch.observe()[0]["time_ms"], rec["time_ms"][N_PER_RANK:])
