"""BASELINE config 5 on CPU: two `gloo` ranks, rank 0 holding player 0 and rank 1 holding player 1 of the same games,
exchanging garbage lines / dead flags by all-gather (drl-tetris_amd/distributed.py SplitOpponents), must reproduce the
co-located two-player game bit for bit: per-step done flags, per-board state incl. garbage queues, last_winner."""
import os
import socket
import sys

import numpy as np
import torch.distributed as dist
import torch.multiprocessing as mp

import __graft_entry__ as ge
from oracle import oracle as orc

N, STEPS = 160, 260
CHECK_EVERY = 20
FIELDS = ["x", "y", "piece", "cur_rot", "next", "dead", "reward", "inc_count", "combo_count", "combo_remaining", "time_ms", "incoming",
          "fifo_len", "fifo_count", "fifo_delay", "min_remaining", "lines_sent", "lines_cleared", "lines_blocked", "piece_draws",
          "hole_draws", "drop_time", "lock_time", "lock_armed", "combo_start", "combo_time"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _actions(step, scenario="random"):
    rng = np.random.default_rng(1000 + step)
    if scenario == "o_only":
        # O pieces dropped into column pairs 0-1, 2-3, ... clear two lines every five pieces: garbage and combo lines
        # flow in both directions all the time; a little noise makes holes so that games also end
        k = (step // 2 + np.arange(N)) % 5
        trans = np.where(rng.random(N) < 0.06, rng.integers(0, 10, N), 2 * k).astype(np.uint8)
        return np.zeros(N, np.uint8), trans, np.full(N, step % 2, np.uint8)
    # a crude but line-clearing policy mix: mostly flat placements so that garbage actually flows
    rot = rng.integers(0, 4, N).astype(np.uint8)
    trans = ((np.arange(N) * 3 + step * 2 + rng.integers(0, 3, N)) % 10).astype(np.uint8)
    acting = np.where(rng.random(N) < 0.15, rng.integers(0, 2, N), step % 2).astype(np.uint8)
    return rot, trans, acting


def _worker(rank, world, port, out_dir, scenario, pieces):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ge.ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mod = __import__("importlib").import_module("drl-tetris_amd.distributed")
    seeds = orc.episode_seed(np.arange(N), 0)
    so = mod.SplitOpponents(N, side=rank, peer=1 - rank, dist=dist, seeds=seeds, pieces=pieces, lib_path=ge.build_harness())
    dones, episode, mid = [], np.zeros(N, np.int64), []
    for s in range(STEPS):
        rot, trans, acting = _actions(s, scenario)
        done, lines, dead = so.step_rt(rot, trans, acting)
        dones.append(done)
        idx = np.nonzero(done)[0].astype(np.int32)
        if len(idx):
            episode[idx] += 1
            so.reset(idx, orc.episode_seed(idx, episode[idx]))
        if s % CHECK_EVERY == CHECK_EVERY - 1:
            mid.append(so.batch.observe()[0])          # state right after this step's resets
    rec, ro, lw = so.batch.observe()
    from tests import engines
    counts = engines.harness_path_counts()
    np.savez(os.path.join(out_dir, f"side{rank}.npz"), rec=rec, ro=ro, lw=lw, dones=np.stack(dones), mid=np.stack(mid),
             undo=np.array([counts["undo_simple"], counts["undo_full"], counts["undo_mispredict"]]))
    so.close()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("scenario,pieces", [("random", (0, 1, 2, 3, 4, 5, 6)), ("o_only", (6,))])
def test_split_opponents_equal_colocated_game(tmp_path, scenario, pieces):
    ge.build_harness()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), scenario, pieces), nprocs=2, join=True)
    ref = orc.OracleBatch(N, 2, 20, 10, pieces=pieces, seeds=orc.episode_seed(np.arange(N), 0))
    episode = np.zeros(N, np.int64)
    want_dones, sent_total, max_queue, max_combo, want_mid = [], 0, 0, 0, []
    for s in range(STEPS):
        rot, trans, acting = _actions(s, scenario)
        d = ref.step_rt(rot, trans, acting)
        want_dones.append(d.copy())
        r = ref.observe()[0]
        max_queue, max_combo = max(max_queue, int(r["fifo_len"].max())), max(max_combo, int(r["max_combo"].max()))
        pending_mid = s % CHECK_EVERY == CHECK_EVERY - 1
        idx = np.nonzero(d)[0].astype(np.int32)
        if len(idx):
            sent_total += int(ref.observe(idx)[0]["lines_sent"].sum())
            episode[idx] += 1
            ref.reset(idx, orc.episode_seed(idx, episode[idx]))
        if pending_mid:
            want_mid.append(ref.observe()[0])
    rec, ro, lw = ref.observe()
    if scenario == "o_only":     # the scenario must really exercise the exchange: queued garbage, combos, many lines
        assert sent_total > 300 and max_queue >= 2 and max_combo >= 2, (sent_total, max_queue, max_combo)
    else:
        assert sent_total > 20
    undo = np.load(os.path.join(str(tmp_path), "side1.npz"))["undo"]
    print("undo paths on side 1 (in place, from the full copy):", undo.tolist())
    assert undo[0] > 0 and undo[2] == 0                    # player 1's speculative pass was taken back at all; "simple" was never mispredicted
    for side in (0, 1):
        got = np.load(os.path.join(str(tmp_path), f"side{side}.npz"))
        assert np.array_equal(got["dones"], np.stack(want_dones)), f"done flags differ on side {side}"
        for f in FIELDS:
            assert np.array_equal(got["rec"][f][:, 0], rec[f][:, side]), (side, f)
        assert np.array_equal(got["rec"]["field"][:, 0] > 0, rec["field"][:, side] > 0)
        assert np.array_equal(got["ro"], ro), side
        assert np.array_equal(got["lw"], lw), side
        for k, w in enumerate(want_mid):               # full state every CHECK_EVERY steps, resets included
            for f in FIELDS:
                assert np.array_equal(got["mid"][k][f][:, 0], w[f][:, side]), (side, f, "checkpoint", k)


STEPS_ROLLOUT = 300


def _rollout_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ge.ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mod = __import__("importlib").import_module("drl-tetris_amd.distributed")
    so = mod.SplitOpponents(N, side=rank, peer=1 - rank, dist=dist, seeds=orc.episode_seed(np.arange(N), 0), lib_path=ge.build_harness())
    so.rollout(STEPS_ROLLOUT // 2)
    so.rollout(STEPS_ROLLOUT - STEPS_ROLLOUT // 2, first_step=STEPS_ROLLOUT // 2)
    rec, ro, lw = so.batch.observe()
    np.savez(os.path.join(out_dir, f"roll{rank}.npz"), rec=rec, ro=ro, lw=lw, totals=so.batch.rollout_totals())
    so.close()
    dist.destroy_process_group()


def test_split_rollout_equals_oracle_two_player_rollout(tmp_path):
    """The device-driven split rollout (policy, acting player and auto-reset computed on the device, identically on both
    ranks) against the oracle's co-located two-player rollout: counters and every board."""
    ge.build_harness()
    mp.spawn(_rollout_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    ref = orc.OracleBatch(N, 2, 20, 10, seeds=orc.episode_seed(np.arange(N), 0))
    _, want = ref.rollout_random(STEPS_ROLLOUT)
    rec, ro, lw = ref.observe()
    got = [np.load(os.path.join(str(tmp_path), f"roll{s}.npz")) for s in (0, 1)]
    episodes, lines, sent = int(want[1]), int(want[2]), int(want[3])
    # totals = {env-steps counted on the device, episodes, lines, sent}; steps and episodes are per game (both sides count them)
    assert int(got[0]["totals"][0]) == int(want[0]) == int(got[1]["totals"][0]) == N * STEPS_ROLLOUT
    assert int(got[0]["totals"][1]) == episodes == int(got[1]["totals"][1]) and episodes > 0
    assert int(got[0]["totals"][2]) + int(got[1]["totals"][2]) == lines
    assert int(got[0]["totals"][3]) + int(got[1]["totals"][3]) == sent
    for side in (0, 1):
        for f in FIELDS:
            assert np.array_equal(got[side]["rec"][f][:, 0], rec[f][:, side]), (side, f)
        assert np.array_equal(got[side]["rec"]["field"][:, 0] > 0, rec["field"][:, side] > 0)
        assert np.array_equal(got[side]["ro"], ro) and np.array_equal(got[side]["lw"], lw)


def _pairs_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ge.ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mod = __import__("importlib").import_module("drl-tetris_amd.distributed")
    pair = rank // 2
    so = mod.SplitOpponents(N, side=rank % 2, peer=rank ^ 1, dist=dist, seeds=mod.episode_seeds(pair * N, N), lib_path=ge.build_harness())
    assert so.group is not None and so.g1.shape == (2, N)      # the exchange runs in the pair's own group and moves the peer's row only
    so.batch.set_game_offset(pair * N)
    so.rollout(STEPS_ROLLOUT // 3)
    so.rollout(STEPS_ROLLOUT - STEPS_ROLLOUT // 3, first_step=STEPS_ROLLOUT // 3)
    rec, ro, lw = so.batch.observe()
    np.savez(os.path.join(out_dir, f"pairs{rank}.npz"), rec=rec, ro=ro, lw=lw, totals=so.batch.rollout_totals())
    so.close()
    dist.destroy_process_group()


def test_two_pairs_exchange_in_pair_local_groups(tmp_path):
    """Four ranks = two pairs (BASELINE config 5's shape at 8 GPUs is four): ranks 2k / 2k + 1 hold player 0 / player 1 of games
    [k N, (k + 1) N) and all-gather with each other only (a process group per pair).  Together they must equal ONE oracle rollout
    of 2 N two-player games."""
    ge.build_harness()
    world = 4
    mp.spawn(_pairs_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    ref = orc.OracleBatch(2 * N, 2, 20, 10, seeds=orc.episode_seed(np.arange(2 * N), 0))
    _, want = ref.rollout_random(STEPS_ROLLOUT)
    rec, ro, lw = ref.observe()
    got = [np.load(os.path.join(str(tmp_path), f"pairs{r}.npz")) for r in range(world)]
    assert sum(int(got[r]["totals"][0]) for r in (0, 2)) == int(want[0]) == 2 * N * STEPS_ROLLOUT
    assert sum(int(got[r]["totals"][1]) for r in (0, 2)) == int(want[1])
    assert sum(int(g["totals"][2]) for g in got) == int(want[2]) and sum(int(g["totals"][3]) for g in got) == int(want[3])
    for r in range(world):
        sl, side = slice((r // 2) * N, (r // 2 + 1) * N), r % 2
        for f in FIELDS:
            assert np.array_equal(got[r]["rec"][f][:, 0], rec[f][sl, side]), (r, f)
        assert np.array_equal(got[r]["rec"]["field"][:, 0] > 0, rec["field"][sl, side] > 0)
        assert np.array_equal(got[r]["ro"], ro[sl]) and np.array_equal(got[r]["lw"], lw[sl])
