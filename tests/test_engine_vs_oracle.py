"""The bitboard engine against the oracle on seeded batches: same seeds, same action stream, full
hidden state compared (incl. timers, garbage FIFO, RNG draw counters).  Sizes are chosen so the
oracle finishes in seconds; the full BASELINE sizes are in test_full_size.py."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import engines


def _pair(kind, n, P, H=20, pieces=(0, 1, 2, 3, 4, 5, 6), seed_base=0):
    seeds = orc.episode_seed(np.arange(n) + seed_base, 0)
    eng = engines.make(kind, n, P, H, pieces, seeds=seeds)
    ref = engines.make("oracle", n, P, H, pieces, seeds=seeds)
    return eng, ref


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
@pytest.mark.parametrize("P,H", [(1, 20), (2, 20), (2, 22)])
def test_step_rt_with_resets(kind, P, H):
    n = 4096 if kind == "hip" else 384
    steps = 160
    eng, ref = _pair(kind, n, P, H)
    engines.assert_same_state(eng, ref, where="after create")
    rng = np.random.default_rng(5 + P)
    episode = np.zeros(n, np.int64)
    for s in range(steps):
        rot = rng.integers(0, 4, n).astype(np.uint8)
        trans = rng.integers(0, 10, n).astype(np.uint8)
        player = rng.integers(0, P, n).astype(np.uint8) if s % 3 == 0 else np.full(n, s % P, np.uint8)
        d1, lines, dead = eng.step_rt(rot, trans, player, full=True)
        d2 = ref.step_rt(rot, trans, player)
        assert np.array_equal(d1, d2), f"done differs at step {s}"
        if s % 16 == 15:
            engines.assert_same_state(eng, ref, where=f"step {s}")
            rec = ref.observe()[0]
            assert np.array_equal(lines, rec["reward"]) and np.array_equal(dead, rec["dead"])
        idx = np.nonzero(d2)[0].astype(np.int32)
        if s % 40 == 39:                      # also reset some running games (last_winner = -1 path)
            idx = np.union1d(idx, np.arange(0, n, 7)).astype(np.int32)
        if len(idx):
            episode[idx] += 1
            sd = orc.episode_seed(idx, episode[idx])
            eng.reset(idx, sd)
            ref.reset(idx, sd)
    engines.assert_same_state(eng, ref, where="end")


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
@pytest.mark.parametrize("P,H", [(1, 20), (2, 20), (1, 9), (1, 30)])
def test_rt_from_arbitrary_positions(kind, P, H):
    """(r, t) actions applied to pieces that are NOT at their spawn position: steps of arbitrary keys without a hard
    drop (slides, soft drops, rotations with kicks; gravity and lock-down move the piece meanwhile) leave pieces mid-air,
    next to walls and under overhangs of messy stacks; the following step_rt must rotate with the reference's kick order,
    slide and hard-drop from there (gameField.cpp:10-103, gamePlay.cpp:48-52).  Exercises the kick loop with several kicks
    per action and the exact hard-drop routine behind the byte-parallel one."""
    n = 2048 if kind == "hip" else 256
    K = 12
    eng, ref = _pair(kind, n, P, H, seed_base=4000)
    if kind == "harness":
        engines.harness_path_counts()              # clear
    rng = np.random.default_rng(100 * P + H)
    episode = np.zeros(n, np.int64)
    for s in range(140):
        if s % 3 != 2:
            # keys 1..6 and 8..10 only: no hard drop, so the piece stays in play wherever the keys and the timers leave it
            pool = np.array([1, 2, 3, 4, 5, 6, 6, 6, 8, 9, 10], np.uint8)
            lens = rng.integers(0, K, (n, P)).astype(np.uint8)
            keys = pool[rng.integers(0, len(pool), (n, P, K))]
            done, _, _ = eng.step_keys(keys, lens)
            ref.make_actions(keys, lens)
            d2 = ref.finish_actions(400)
        else:
            rot = rng.integers(0, 4, n).astype(np.uint8)
            trans = rng.integers(0, 10, n).astype(np.uint8)
            player = rng.integers(0, P, n).astype(np.uint8)
            done, _, _ = eng.step_rt(rot, trans, player, full=True)
            d2 = ref.step_rt(rot, trans, player)
        assert np.array_equal(done, d2), f"done differs at step {s}"
        if s % 12 == 11:
            engines.assert_same_state(eng, ref, where=f"step {s}")
        idx = np.nonzero(d2)[0].astype(np.int32)
        if len(idx):
            episode[idx] += 1
            sd = orc.episode_seed(idx + 4000, episode[idx])
            eng.reset(idx, sd)
            ref.reset(idx, sd)
    engines.assert_same_state(eng, ref, where="end")
    if kind == "harness":                          # the run really reached the rare paths it is here for
        counts = engines.harness_path_counts()
        for name in ("rt_off_spawn", "kick", "kick_2nd", "kick_failed", "kick_down", "drop_exact"):
            assert counts[name] > 0, (name, counts)


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
@pytest.mark.parametrize("P", [1, 2])
def test_random_key_sequences(kind, P):
    """General key interpreter (PythonHandle.cpp:73-112): every key 0..10, soft drops, keys after a lock,
    empty lists, make/finish called separately and fused, on subsets of games."""
    n = 2048 if kind == "hip" else 256
    K = 20
    eng, ref = _pair(kind, n, P, 20, seed_base=1000)
    rng = np.random.default_rng(11)
    episode = np.zeros(n, np.int64)
    for s in range(120):
        lens = rng.integers(0, K - 2, (n, P)).astype(np.uint8)
        keys = rng.integers(0, 11, (n, P, K)).astype(np.uint8)
        lock = rng.random((n, P)) < 0.7
        for p in range(P):
            rows = np.nonzero(lock[:, p])[0]
            keys[rows, p, lens[rows, p]] = 7
            lens[rows, p] += 1
        if P == 2:                                 # usually only one player acts (the other gets [0])
            idle = rng.integers(0, 2, n)
            solo = rng.random(n) < 0.8
            for p in range(2):
                rows = np.nonzero(solo & (idle == p))[0]
                keys[rows, p, 0] = 0
                lens[rows, p] = 1
        if s % 2 == 0:
            done, _, _ = eng.step_keys(keys, lens)
        else:
            sub = np.sort(rng.choice(n, n // 2, replace=False)).astype(np.int32)
            rest = np.setdiff1d(np.arange(n), sub).astype(np.int32)
            eng.make_actions(keys[sub], lens[sub], idx=sub)
            eng.make_actions(keys[rest], lens[rest], idx=rest)
            done = eng.finish_actions(400)
        ref.make_actions(keys, lens)
        d2 = ref.finish_actions(400)
        assert np.array_equal(done, d2), f"done differs at step {s}"
        if s % 10 == 9:
            engines.assert_same_state(eng, ref, where=f"step {s}")
        idx = np.nonzero(d2)[0].astype(np.int32)
        if len(idx):
            episode[idx] += 1
            sd = orc.episode_seed(idx + 1000, episode[idx])
            eng.reset(idx, sd)
            ref.reset(idx, sd)
    engines.assert_same_state(eng, ref, where="end")


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
def test_simulate_without_finalize_and_snapshot_restore(kind):
    """tetris_environment.simulate_actions (tetris_environment.py:87-100): copy, make_action without
    finish_action, look, set back.  Also copy/set across games and batches (PythonHandle.cpp:36-42)."""
    n, P = 128, 2
    eng, ref = _pair(kind, n, P)
    rng = np.random.default_rng(3)
    for s in range(30):
        rot, trans = rng.integers(0, 4, n).astype(np.uint8), rng.integers(0, 10, n).astype(np.uint8)
        eng.step_rt(rot, trans, s % 2)
        ref.step_rt(rot, trans, s % 2)
    anchor = eng.snapshot()
    ref_anchor = engines.make("oracle", n, P)
    ref_anchor.copy_from(ref)
    keys = np.zeros((n, P, 8), np.uint8)
    lens = np.ones((n, P), np.uint8)
    keys[:, 0, :4] = [8, 2, 3, 7]
    lens[:, 0] = 4
    eng.make_actions(keys, lens)
    ref.make_actions(keys, lens)
    engines.assert_same_state(eng, ref, where="after make without finish")
    eng.restore(anchor)
    ref.copy_from(ref_anchor)
    engines.assert_same_state(eng, ref, where="after restore")
    # restore game 5's snapshot into games 0..9 of a second batch, then play on: RNG position travels
    other = engines.make(kind, 16, P)
    other_ref = engines.make("oracle", 16, P)
    other.restore(np.repeat(anchor[5:6], 10, axis=0), idx=np.arange(10, dtype=np.int32))
    other_ref.copy_from(ref_anchor, dst_idx=np.arange(10), src_idx=np.full(10, 5))
    for s in range(40):
        rot, trans = rng.integers(0, 4, 16).astype(np.uint8), rng.integers(0, 10, 16).astype(np.uint8)
        other.step_rt(rot, trans, s % 2)
        other_ref.step_rt(rot, trans, s % 2)
    engines.assert_same_state(other, other_ref, where="restored games played on")
    # state.lock(): Python writes dead = 1 into a snapshot's players (data_types/state.py:9-12)
    dead = np.ones((n, P), np.uint8)
    eng.set_dead(dead)
    ref.set_dead(dead)
    eng.step_rt(rot[:1].repeat(n), trans[:1].repeat(n), 0)
    ref.step_rt(rot[:1].repeat(n), trans[:1].repeat(n), 0)
    engines.assert_same_state(eng, ref, where="locked state does not move")


class _DevArrays:
    """uint8 / int16 arrays the `_dev` entry points can take: numpy for the CPU harness (its "device" is host memory),
    torch tensors on the GPU."""

    def __init__(self, kind):
        self.gpu = kind == "hip"
        if self.gpu:
            import torch
            self.torch = torch

    def put(self, a):
        return self.torch.as_tensor(np.ascontiguousarray(a)).cuda() if self.gpu else np.ascontiguousarray(a).copy()

    def ptr(self, t):
        import ctypes as C
        if t is None:
            return None
        return C.c_void_p(t.data_ptr()) if self.gpu else t.ctypes.data_as(C.c_void_p)

    def get(self, t):
        return t.cpu().numpy() if self.gpu else t.copy()


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
@pytest.mark.parametrize("P", [1, 2])
def test_device_side_auto_reset_and_masked_reset(kind, P):
    """tetris_step_rt_dev_ex(TETRIS_STEP_AUTO_RESET) and tetris_reset_dev against the oracle driven the way the reference's
    worker loop drives its envs (worker.py:157-166: step, then reset the finished ones): 200 steps, the outputs of every step
    (done / lines / dead BEFORE the reset) and the complete state; seeds follow the built-in schedule by game and episode."""
    n = 4096 if kind == "hip" else 320
    eng, ref = _pair(kind, n, P)
    eng.set_game_offset(1000)                                  # the schedule is keyed by GLOBAL game id
    D = _DevArrays(kind)
    rng = np.random.default_rng(11 + P)
    episode = np.zeros(n, np.int64)
    done_d, lines_d, dead_d = D.put(np.zeros(n, np.uint8)), D.put(np.zeros((P, n), np.uint8)), D.put(np.zeros((P, n), np.uint8))
    total_done = 0
    for s in range(200):
        rot, trans = rng.integers(0, 4, n).astype(np.uint8), rng.integers(0, 10, n).astype(np.uint8)
        player = rng.integers(0, P, n).astype(np.uint8)
        auto = s % 50 < 40                                     # mostly auto-reset; sometimes a separate masked reset launch
        r_d, t_d, p_d = D.put(rot), D.put(trans), D.put(player)
        eng.step_rt_dev(D.ptr(r_d), D.ptr(t_d), D.ptr(p_d), D.ptr(done_d), D.ptr(lines_d), D.ptr(dead_d), auto_reset=auto)
        if not auto:
            eng.reset_dev(D.ptr(done_d), None)                 # mask = this step's done flags, seeds from the schedule
        d_ref = ref.step_rt(rot, trans, player)
        rec = ref.observe()[0]
        if kind == "hip":
            eng.sync()
        assert np.array_equal(D.get(done_d), d_ref), s
        assert np.array_equal(D.get(lines_d).T, rec["reward"]) and np.array_equal(D.get(dead_d).T, rec["dead"]), s
        idx = np.nonzero(d_ref)[0].astype(np.int32)
        total_done += len(idx)
        if len(idx):
            episode[idx] += 1
            ref.reset(idx, orc.episode_seed(idx + 1000, episode[idx]))
        if s % 25 == 24:
            engines.assert_same_state(eng, ref, where=f"step {s}")
    assert total_done > n                                      # every game was reset on the device more than once on average
    # explicit device seeds for a masked subset
    mask = (np.arange(n) % 3 == 0).astype(np.uint8)
    seeds = rng.integers(-32768, 32767, n).astype(np.int16)
    m_d, s_d = D.put(mask), D.put(seeds)
    eng.reset_dev(D.ptr(m_d), D.ptr(s_d))
    idx = np.nonzero(mask)[0].astype(np.int32)
    ref.reset(idx, seeds[idx])
    engines.assert_same_state(eng, ref, where="masked reset with device seeds")


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
@pytest.mark.parametrize("P", [3, 4])
def test_three_and_four_players(kind, P):
    """More than two players per game (PythonHandle(n_players, ...), PythonHandle.cpp:5-25; distributeLines gives every opponent
    amount / (P - 1) lines as a float, :124-136; the round ends when fewer than two players are alive): (r, t) steps and raw key
    lists for random players with resets, enumeration and get_actions for the last player, snapshot / restore, the built-in
    rollout — all against the oracle, which the reference's own 3- and 4-player traces pin (tests/golden/trace_*_3p / _4p)."""
    n = 2048 if kind == "hip" else 160
    seeds = orc.episode_seed(np.arange(n), 0)
    eng, ref = engines.make(kind, n, P, 20, (6, 4), seeds=seeds), engines.make("oracle", n, P, 20, (6, 4), seeds=seeds)
    rng = np.random.default_rng(100 + P)
    episode = np.zeros(n, np.int64)
    blob, blob_ref, total_done = None, None, 0
    for s in range(300):
        player = rng.integers(0, P, n).astype(np.uint8)
        if s % 3 == 2:                                         # raw key lists for every player of every game
            lens = rng.integers(0, 6, (n, P)).astype(np.uint8)
            keys = rng.integers(0, 11, (n, P, 6)).astype(np.uint8)
            keys[np.arange(n), player, np.maximum(lens[np.arange(n), player], 1) - 1] = 7
            lens[np.arange(n), player] = np.maximum(lens[np.arange(n), player], 1)
            d1 = eng.step_keys(keys, lens)[0]
            ref.make_actions(keys, lens)
            d2 = ref.finish_actions(400)
        else:                                                  # O and I pieces laid side by side: lines, combos, garbage thirds / halves
            trans = np.where(rng.random(n) < 0.1, rng.integers(0, 10, n), 2 * ((s // P + np.arange(n)) % 5)).astype(np.uint8)
            rot = (rng.random(n) < 0.1).astype(np.uint8)
            d1, d2 = eng.step_rt(rot, trans, player), ref.step_rt(rot, trans, player)
        assert np.array_equal(d1, d2), s
        if s == 120:
            blob, blob_ref = eng.snapshot(), engines.make("oracle", n, P, 20, (6, 4))
            blob_ref.copy_from(ref)
        idx = np.nonzero(d2)[0].astype(np.int32)
        total_done += len(idx)
        if len(idx):
            episode[idx] += 1
            sd = orc.episode_seed(idx, episode[idx])
            eng.reset(idx, sd); ref.reset(idx, sd)
        if s % 60 == 59:
            engines.assert_same_state(eng, ref, where=f"step {s}")
    rec = ref.observe()[0]
    assert rec["lines_sent"].sum() > 0 and (rec["incoming"] % 1 != 0).any() or rec["lines_blocked"].sum() > 0     # fractional garbage was in play
    last = np.full(n, P - 1, np.uint8)
    v1, y1, c1, a1 = eng.enumerate_drops(player=last)
    v2, y2, c2, a2 = ref.enumerate_drops(player=last)
    ok = v2.astype(bool)
    assert np.array_equal(v1, v2) and np.array_equal(y1[ok], y2[ok]) and np.array_equal(c1[ok], c2[ok])
    sub = np.arange(0, n, 37, dtype=np.int32)
    assert eng.get_actions(sub, player=P - 1) == [ref.get_actions(int(g), P - 1) for g in sub]
    with pytest.raises(Exception):
        eng.observe_packed(player=0)                           # own / opponent planes: two players at most
    eng.restore(blob); ref.copy_from(blob_ref)
    engines.assert_same_state(eng, ref, where="restored at step 120")
    c_gpu, _ = eng.rollout_random(40, 2, first_step=7)
    _, c_ref = ref.rollout_random(80, first_step=7)
    assert c_gpu.tolist() == c_ref.tolist()
    engines.assert_same_state(eng, ref, where="after the built-in rollout")
    assert total_done > 0


def _expected_packed(rec, me, P, H):
    """state_dict + unpacker semantics (state_processors.py:23-54; state_unpack.py:88-137) computed from oracle records:
    -> visual [P][n][H][10], vector [P][n][12], piece [P][n], slot 0 = player me[i]'s board"""
    n = len(me)
    visual, vector, piece = np.zeros((P, n, H, 10), np.uint8), np.zeros((P, n, 12), np.uint8), np.zeros((P, n), np.uint8)
    for sl in range(P):
        who = me if sl == 0 else (P - 1 - me)
        r = rec[np.arange(n), who]
        visual[sl] = (r["field"][:, :H, :] > 0).astype(np.uint8)
        vector[sl, :, 0] = r["x"].astype(np.uint8); vector[sl, :, 1] = r["y"].astype(np.uint8); vector[sl, :, 2] = r["inc_count"]
        vector[sl, :, 3] = np.minimum(25000, (r["combo_remaining"].astype(np.uint32) + 50) & 0xFFFF) // 100
        vector[sl, :, 4] = r["combo_count"]
        vector[sl, np.arange(n), 5 + r["next"]] = 1
        piece[sl] = r["piece"]
    return visual, vector, piece


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
@pytest.mark.parametrize("P,H", [(1, 20), (2, 20), (2, 22), (1, 21), (2, 30)])
def test_step_and_observation_in_one_launch(kind, P, H):
    """tetris_step_rt_observe_dev = one iteration of the agent loop in one launch (worker.py:91-118: perform_action, then
    get_state + unpack): against the oracle's step (done / lines / dead, auto-reset with the schedule's seeds) and against the
    observation the oracle's state gives for `next_player` (state_processors.py:23-54, state_unpack.py:88-137), every step;
    n is not a multiple of the 64 games a wave holds, H = 21 runs the two-kernel fallback (odd height), H = 30 needs more
    than 48 KB of dynamic LDS per workgroup in the two-player kernel."""
    n = 4096 + 37 if kind == "hip" else 200 + 37
    seeds = orc.episode_seed(np.arange(n), 0)
    eng, ref = engines.make(kind, n, P, H, seeds=seeds), engines.make("oracle", n, P, H, seeds=seeds)
    D = _DevArrays(kind)
    rng = np.random.default_rng(5 + 10 * P + H)
    episode = np.zeros(n, np.int64)
    done_d, lines_d, dead_d = D.put(np.zeros(n, np.uint8)), D.put(np.zeros((P, n), np.uint8)), D.put(np.zeros((P, n), np.uint8))
    vis_d, vec_d, pc_d = D.put(np.zeros((P, n, H, 10), np.uint8)), D.put(np.zeros((P, n, 12), np.uint8)), D.put(np.zeros((P, n), np.uint8))
    total_done = 0
    for s in range(120):
        rot, trans = rng.integers(0, 4, n).astype(np.uint8), rng.integers(0, 10, n).astype(np.uint8)
        player, nxt = rng.integers(0, P, n).astype(np.uint8), rng.integers(0, P, n).astype(np.uint8)
        auto = s % 10 != 9
        r_d, t_d, p_d, n_d = D.put(rot), D.put(trans), D.put(player), D.put(nxt)
        eng.step_rt_observe_dev(D.ptr(r_d), D.ptr(t_d), D.ptr(p_d), D.ptr(done_d), D.ptr(lines_d), D.ptr(dead_d),
                                D.ptr(n_d) if s % 7 else (None if P == 1 or not nxt.any() else D.ptr(n_d)),
                                D.ptr(vis_d), D.ptr(vec_d), D.ptr(pc_d), auto_reset=auto)
        d_ref = ref.step_rt(rot, trans, player)
        rec = ref.observe()[0]
        if kind == "hip":
            eng.sync()
        assert np.array_equal(D.get(done_d), d_ref), s
        assert np.array_equal(D.get(lines_d).T, rec["reward"]) and np.array_equal(D.get(dead_d).T, rec["dead"]), s
        idx = np.nonzero(d_ref)[0].astype(np.int32)
        total_done += len(idx)
        if len(idx) and auto:
            episode[idx] += 1
            ref.reset(idx, orc.episode_seed(idx, episode[idx]))
        me = nxt if (s % 7 or (P > 1 and nxt.any())) else np.zeros(n, np.uint8)      # a NULL next_player means player 0
        want_vis, want_vec, want_pc = _expected_packed(ref.observe()[0], me, P, H)
        assert np.array_equal(D.get(pc_d), want_pc), s
        assert np.array_equal(D.get(vec_d), want_vec), s
        assert np.array_equal(D.get(vis_d), want_vis), s
        if not auto:                                           # the observation above showed the finished games as they ended
            eng.reset_dev(D.ptr(done_d), None)                 # mask = this step's done flags, seeds from the schedule
            if len(idx):
                episode[idx] += 1
                ref.reset(idx, orc.episode_seed(idx, episode[idx]))
    assert total_done > n // 4
    engines.assert_same_state(eng, ref, where="after the fused loop")


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
def test_device_driven_loop_outlives_the_resident_rng_tables(kind):
    """An asynchronous `_dev` loop whose episodes outlive the resident RNG-table chunks (2 x 624 draws): O pieces laid side
    by side clear two rows every five pieces, so the games never end.  Nothing in the loop synchronises; the request to
    extend the tables reaches the host through the batch's flag words (include/tetris_hip.h) and must be answered before any
    board runs past the tables."""
    n, P, steps = 256, 1, 1500
    seeds = orc.episode_seed(np.arange(n), 0)
    eng = engines.make(kind, n, P, 20, (6,), seeds=seeds)
    ref = engines.make("oracle", n, P, 20, (6,), seeds=seeds)
    D = _DevArrays(kind)
    rot = np.zeros(n, np.uint8)
    r_d = D.put(rot)
    t_ds = [D.put(np.full(n, 2 * k, np.uint8)) for k in range(5)]
    done_d = D.put(np.zeros(n, np.uint8))
    for s in range(steps):
        eng.step_rt_dev(D.ptr(r_d), D.ptr(t_ds[s % 5]), None, D.ptr(done_d), None, None, auto_reset=True)
        d = ref.step_rt(rot, np.full(n, 2 * (s % 5), np.uint8))
        assert not d.any()
    eng.sync()
    assert eng.table_chunks >= 3
    engines.assert_same_state(eng, ref, where="end")
    assert int(ref.observe()[0]["piece_draws"].min()) > 1248


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
def test_rng_tables_extend_past_two_chunks(kind):
    """An episode that outlives the resident RNG-table chunks (624 draws each): key [1] never locks, so
    every finish_action deals one piece."""
    n, P = 4, 1
    eng, ref = _pair(kind, n, P, seed_base=31000)
    keys = np.ones((n, P, 1), np.uint8)
    lens = np.ones((n, P), np.uint8)
    for s in range(1400):
        eng.step_keys(keys, lens)
        ref.make_actions(keys, lens)
        ref.finish_actions(400)
        if s % 100 == 99 or s > 1180:
            a, b = eng.observe()[0], ref.observe()[0]
            assert np.array_equal(a["next"], b["next"]) and np.array_equal(a["piece_draws"], b["piece_draws"]), s
    assert eng.table_chunks >= 3            # tables are shared per process: another test may have grown them already
    engines.assert_same_state(eng, ref, where="end")


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
@pytest.mark.parametrize("P,H,pieces", [(1, 20, (4,)), (1, 12, (2, 3)), (2, 20, (0, 1)), (2, 16, (5, 6, 4)), (1, 31, (0, 1, 2, 3, 4, 5, 6)),
                                        (1, 4, (6,))])
def test_rollout_piece_sets_and_heights(kind, P, H, pieces):
    """Built-in rollouts for restricted piece sets (tetris_env.set_pieces: the piece map is cycled over the 7 slots, only
    S/Z keeps the redraw loop open, gamePlay.cpp:218-230) and unusual heights, fused and unfused, against the oracle."""
    n = 1024 if kind == "hip" else 128
    seeds = orc.episode_seed(np.arange(n), 0)
    eng = engines.make(kind, n, P, H, pieces, seeds=seeds)
    eng2 = engines.make(kind, n, P, H, pieces, seeds=seeds)
    ref = engines.make("oracle", n, P, H, pieces, seeds=seeds)
    c1, _ = eng.rollout_random(4, 30)
    c2, _ = eng2.rollout_random(120, 1)
    _, c3 = ref.rollout_random(120)
    assert c1.tolist() == c2.tolist() == c3.tolist()
    engines.assert_same_state(eng, ref, where="fused")
    engines.assert_same_state(eng2, ref, where="unfused")


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
@pytest.mark.parametrize("P", [1, 2])
def test_rollout_random_matches_oracle(kind, P):
    """The built-in synthetic rollout (SURVEY §8d policy + auto-reset) against the oracle's: counters and
    final state, fused (many steps per launch) and unfused (one step per launch) give the same result."""
    n = 2048 if kind == "hip" else 192
    seeds = orc.episode_seed(np.arange(n), 0)
    eng = engines.make(kind, n, P, seeds=seeds)
    eng2 = engines.make(kind, n, P, seeds=seeds)
    ref = engines.make("oracle", n, P, seeds=seeds)
    c1, _ = eng.rollout_random(6, 32)               # 192 steps, fused
    c2a, _ = eng2.rollout_random(100, 1)            # 100 + 92 steps, one per launch
    c2b, _ = eng2.rollout_random(92, 1, first_step=100)
    ep, c3 = ref.rollout_random(192)
    assert c1.tolist() == c3.tolist() == (c2a + c2b).tolist()
    assert int(c1[0]) == n * 192 and int(c1[1]) > 0
    engines.assert_same_state(eng, ref, where="fused rollout")
    engines.assert_same_state(eng2, ref, where="unfused rollout")


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
def test_enumerate_drops_matches_oracle(kind):
    """BASELINE config 4 (SIXten-style enumeration): all rotation x column drop afterstates of the current piece
    (TestField.cpp:64-125), on boards taken from mid-game, both players, against the oracle's cell-based version."""
    n, P = (4096 if kind == "hip" else 256), 2
    eng, ref = _pair(kind, n, P, seed_base=500)
    rng = np.random.default_rng(9)
    for s in range(24):
        rot, trans = rng.integers(0, 4, n).astype(np.uint8), rng.integers(0, 10, n).astype(np.uint8)
        eng.step_rt(rot, trans, s % 2)
        ref.step_rt(rot, trans, s % 2)
        if s in (11, 12, 23):
            player = rng.integers(0, 2, n).astype(np.uint8)
            v1, y1, c1, a1 = eng.enumerate_drops(player=player)
            v2, y2, c2, a2 = ref.enumerate_drops(player=player)
            assert np.array_equal(v1, v2) and np.array_equal(y1, y2) and np.array_equal(c1, c2)
            # columns -> cells: bit y of column c = cell (y, c)
            cells = ((a1[..., None, :] >> np.arange(20, dtype=np.uint32)[:, None]) & 1).astype(bool)      # [n,4,10,20,10]
            assert np.array_equal(cells, a2 > 0)
            assert v1.any(axis=(1, 2)).mean() > 0.9 and c1.max() <= 4
    sub = np.arange(0, n, 3, dtype=np.int32)
    v1, y1, c1, _ = eng.enumerate_drops(idx=sub, player=1, columns=False)
    v2, y2, c2, _ = ref.enumerate_drops(idx=sub, player=1, cells=False)
    assert np.array_equal(v1, v2) and np.array_equal(y1, y2) and np.array_equal(c1, c2)
    # device-pointer form, both output layouts (game-major and rotation-minor column planes), with a ragged last workgroup
    D = _DevArrays(kind)
    m = n - 3
    player = rng.integers(0, 2, m).astype(np.uint8)
    want = eng.enumerate_drops(idx=np.arange(m, dtype=np.int32), player=player)
    for planar in (False, True):
        v_d, y_d, c_d = D.put(np.zeros(m * 40, np.uint8)), D.put(np.zeros(m * 40, np.int8)), D.put(np.zeros(m * 40, np.uint8))
        a_d, p_d = D.put(np.zeros(m * 400, np.uint32).view(np.int32)), D.put(player)
        eng.enumerate_drops_dev(m, D.ptr(v_d), D.ptr(y_d), D.ptr(c_d), D.ptr(a_d), player=D.ptr(p_d), planar=planar)
        eng.sync()
        shape3 = lambda t: t.reshape(m, 10, 4).transpose(0, 2, 1) if planar else t.reshape(m, 4, 10)
        assert np.array_equal(shape3(D.get(v_d)), want[0]) and np.array_equal(shape3(D.get(y_d)), want[1])
        assert np.array_equal(shape3(D.get(c_d)), want[2])
        a = D.get(a_d).view(np.uint32)
        a = a.reshape(10, m, 10, 4).transpose(1, 3, 2, 0) if planar else a.reshape(m, 4, 10, 10)
        assert np.array_equal(a, want[3]), planar


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
def test_get_actions_matches_oracle(kind):
    """Exact ordered key lists of PythonHandle.get_actions (TestField.cpp:64-415) on batches of mid-game boards,
    including boards with overhangs (the games are played with enumerated tuck/spin actions)."""
    n, P = (1024 if kind == "hip" else 96), 2
    eng, ref = _pair(kind, n, P, seed_base=7000)
    rng = np.random.default_rng(21)
    K = 48
    tucks = 0
    for s in range(14):
        player = np.full(n, s % 2, np.uint8)
        lists = eng.get_actions(player=player)
        for g in range(0, n, max(1, n // 48)):
            want = ref.get_actions(g, s % 2)
            assert lists[g] == want, (s, g)
        tucks += sum(1 for l in lists for a in l if 5 in a or 6 in a)
        keys = np.zeros((n, P, K), np.uint8)
        lens = np.ones((n, P), np.uint8)
        for g in range(n):
            a = lists[g][int(rng.integers(len(lists[g])))]
            keys[g, s % 2, : len(a)] = a
            lens[g, s % 2] = len(a)
        d1, _, _ = eng.step_keys(keys, lens)
        ref.make_actions(keys, lens)
        d2 = ref.finish_actions(400)
        assert np.array_equal(d1, d2)
    assert tucks > 0
    engines.assert_same_state(eng, ref, where="after playing enumerated actions")
    sub = np.array([3, 1, 7], np.int32)
    assert eng.get_actions(sub, player=[1, 0, 1]) == [ref.get_actions(3, 1), ref.get_actions(1, 0), ref.get_actions(7, 1)]


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
@pytest.mark.parametrize("P,H", [(2, 20), (1, 22), (2, 21), (1, 31)])
def test_observe_packed_matches_state_dict(kind, P, H):
    """The observation kernel against state_dict + unpacker semantics computed from oracle records
    (state_processors.py:23-54; state_unpack.py:88-137), from each player's perspective."""
    n = 3000 if kind == "hip" else 300
    eng, ref = _pair(kind, n, P, H, seed_base=99)
    rng = np.random.default_rng(2)
    for s in range(40):
        rot, trans = rng.integers(0, 4, n).astype(np.uint8), rng.integers(0, 10, n).astype(np.uint8)
        eng.step_rt(rot, trans, s % P)
        ref.step_rt(rot, trans, s % P)
    rec = ref.observe()[0]
    for trial in range(2):
        me = rng.integers(0, P, n).astype(np.uint8)
        idx = None if trial == 0 else np.sort(rng.choice(n, n // 3, replace=False)).astype(np.int32)
        sel = np.arange(n) if idx is None else idx
        visual, vector, piece = eng.observe_packed(idx, me[sel])
        assert visual.shape == (P, len(sel), H, 10) and vector.shape == (P, len(sel), 12)
        for sl in range(P):
            who = me[sel] if sl == 0 else (P - 1 - me[sel])
            r = rec[sel, who]
            assert np.array_equal(visual[sl], (r["field"][:, :H, :] > 0).astype(np.uint8))
            want = np.zeros((len(sel), 12), np.uint8)
            want[:, 0] = r["x"].astype(np.uint8); want[:, 1] = r["y"].astype(np.uint8); want[:, 2] = r["inc_count"]
            want[:, 3] = np.minimum(25000, (r["combo_remaining"].astype(np.uint32) + 50) & 0xFFFF) // 100
            want[:, 4] = r["combo_count"]
            want[np.arange(len(sel)), 5 + r["next"]] = 1
            assert np.array_equal(vector[sl], want)
            assert np.array_equal(piece[sl], r["piece"])


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
def test_colour_batch_matches_oracle_cells(kind):
    """Colour-tracking batches (TETRIS_FLAG_COLOURS) against the oracle's cell values on a garbage-heavy 2-player batch
    (O pieces filling column pairs), incl. snapshot/restore of the wider state and the garbageCleared statistic."""
    n, P = (2048 if kind == "hip" else 128), 2
    seeds = orc.episode_seed(np.arange(n), 0)
    eng = engines.make(kind, n, P, 20, pieces=(6, 4), seeds=seeds, colours=True)
    ref = engines.make("oracle", n, P, 20, pieces=(6, 4), seeds=seeds)
    rng = np.random.default_rng(4)
    episode = np.zeros(n, np.int64)
    blob, seen_garbage, max_gc = None, False, 0
    for s in range(220):
        k = (s // 2 + np.arange(n)) % 5
        trans = np.where(rng.random(n) < 0.05, rng.integers(0, 10, n), 2 * k).astype(np.uint8)
        rot = (rng.random(n) < 0.1).astype(np.uint8)
        d1 = eng.step_rt(rot, trans, s % 2)
        d2 = ref.step_rt(rot, trans, s % 2)
        assert np.array_equal(d1, d2)
        if s % 20 == 19:
            a, b = eng.observe()[0], ref.observe()[0]
            assert np.array_equal(a["field"], b["field"]), s            # exact cell values 0..8
            assert np.array_equal(a["garbage_cleared"], b["garbage_cleared"])
            seen_garbage |= bool((b["field"] == 8).any())
            max_gc = max(max_gc, int(b["garbage_cleared"].max()))
        if s == 100:
            blob = eng.snapshot()
            assert blob.shape[1] == 5 + P * 69
        idx = np.nonzero(d2)[0].astype(np.int32)
        if len(idx):
            episode[idx] += 1
            eng.reset(idx, orc.episode_seed(idx, episode[idx]))
            ref.reset(idx, orc.episode_seed(idx, episode[idx]))
    assert seen_garbage, "the scenario must push garbage rows"      # cleared garbage rows: golden traces greedy_2p*
    engines.assert_same_state(eng, ref, where="end")
    eng.restore(blob)
    assert np.array_equal(eng.snapshot(), blob)
