"""Edge cases of the batched API against the oracle: empty index lists, a single game, ragged key lists (0..40 keys, the
other player idle or acting too), the smallest and the tallest supported boards, duplicate-free subsets in arbitrary order."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests import engines


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
def test_empty_and_single(kind):
    eng = engines.make(kind, 1, 2, seeds=5)
    ref = engines.make("oracle", 1, 2, seeds=5)
    empty = np.zeros(0, np.int32)
    eng.reset(empty, seeds=0)
    assert eng.finish_actions(400, idx=empty).shape == (0,)
    d, l, dd = eng.step_keys(np.zeros((0, 2, 4), np.uint8), np.zeros((0, 2), np.uint8), idx=empty)
    assert d.shape == (0,) and l.shape == (0, 2)
    assert eng.observe(empty)[0].shape == (0, 2)
    assert eng.snapshot(empty).shape[0] == 0
    assert eng.get_actions(empty) == []
    v, y, c, a = eng.enumerate_drops(empty)
    assert v.shape == (0, 4, 10)
    engines.assert_same_state(eng, ref, where="untouched by empty calls")
    for s in range(60):                     # one game, both players, until it ends and beyond (round_over: nothing moves)
        keys = np.zeros((1, 2, 3), np.uint8); lens = np.ones((1, 2), np.uint8)
        keys[0, s % 2] = [2, 3, 7]; lens[0, s % 2] = 3
        d1, _, _ = eng.step_keys(keys, lens)
        ref.make_actions(keys, lens)
        assert d1[0] == ref.finish_actions(400)[0]
    engines.assert_same_state(eng, ref, where="single game")


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
@pytest.mark.parametrize("H", [5, 31])
def test_extreme_heights(kind, H):
    n, P = 200, 2
    seeds = orc.episode_seed(np.arange(n), 0)
    eng = engines.make(kind, n, P, H, seeds=seeds)
    ref = engines.make("oracle", n, P, H, seeds=seeds)
    rng = np.random.default_rng(H)
    episode = np.zeros(n, np.int64)
    for s in range(150):
        rot, trans = rng.integers(0, 4, n).astype(np.uint8), rng.integers(0, 10, n).astype(np.uint8)
        d1 = eng.step_rt(rot, trans, s % 2)
        d2 = ref.step_rt(rot, trans, s % 2)
        assert np.array_equal(d1, d2), s
        idx = np.nonzero(d2)[0].astype(np.int32)
        if len(idx):
            episode[idx] += 1
            eng.reset(idx, orc.episode_seed(idx, episode[idx]))
            ref.reset(idx, orc.episode_seed(idx, episode[idx]))
    engines.assert_same_state(eng, ref, where=f"H={H}")
    for g in (0, 7, 150):
        assert eng.get_actions(g, 0) == ref.get_actions(g, 0)
    v1, y1, c1, _ = eng.enumerate_drops(player=1, columns=False)
    v2, y2, c2, _ = ref.enumerate_drops(player=1, cells=False)
    assert np.array_equal(v1, v2) and np.array_equal(y1, y2) and np.array_equal(c1, c2)


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
def test_ragged_long_key_lists_on_shuffled_subsets(kind):
    n, P, K = 300, 2, 40
    seeds = orc.episode_seed(np.arange(n), 3)
    eng = engines.make(kind, n, P, 22, seeds=seeds)
    ref = engines.make("oracle", n, P, 22, seeds=seeds)
    rng = np.random.default_rng(8)
    for s in range(50):
        sub = rng.permutation(n)[: rng.integers(1, n)].astype(np.int32)        # unsorted, unique
        m = len(sub)
        lens = rng.integers(0, K + 1, (m, P)).astype(np.uint8)
        lens[rng.random((m, P)) < 0.3] = 0                                      # empty lists are legal
        keys = rng.integers(0, 11, (m, P, K)).astype(np.uint8)
        d1, _, _ = eng.step_keys(keys, lens, idx=sub)
        ref.make_actions(keys, lens, idx=sub)
        d2 = ref.finish_actions(400, idx=sub)
        assert np.array_equal(d1, d2), s
        if s % 10 == 9:
            engines.assert_same_state(eng, ref, where=f"step {s}")
        done = sub[np.nonzero(d2)[0]]
        if len(done):
            eng.reset(done, 1000 + s)
            ref.reset(done, 1000 + s)
    engines.assert_same_state(eng, ref, where="end")


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
def test_garbage_queue_overflow_ends_only_the_offending_game(kind):
    """The reference's garbage queue is an unbounded vector (Garbage.h:27); a board here holds 8 pending packets.  Games 0..n/2
    are driven past that (10 ms ticks, so nothing is released; player 0 clears two rows of O pieces every fifth piece and
    each clear becomes one more packet in player 1's queue), the other half plays the same batch calls with 400 ms ticks'
    worth of ordinary moves.  Up to the overflow every game equals the oracle; the 9th packet ends THAT game's round with
    the error bit set on the board — and nothing else in the batch: no call fails, the other games stay bit-exact, and
    after a reset the flagged games play on like the oracle again."""
    n, P = 64, 2
    half = n // 2
    seeds = orc.episode_seed(np.arange(n), 0)
    eng = engines.make(kind, n, P, 20, (6,), seeds=seeds)
    ref = engines.make("oracle", n, P, 20, (6,), seeds=seeds)
    rng = np.random.default_rng(3)
    flagged_at = None
    for s in range(70):
        keys = np.zeros((n, P, 16), np.uint8)
        lens = np.ones((n, P), np.uint8)
        k = s % 5
        flood = [2] + [3] * (2 * k) + [7]                        # O pieces side by side: two full rows every fifth piece
        keys[:half, 0, : len(flood)] = flood
        lens[:half, 0] = len(flood)
        for g in range(half, n):                                 # the other half: ordinary random (r, t) moves, players alternating
            a = [8] * int(rng.integers(0, 4)) + [2] + [3] * int(rng.integers(0, 10)) + [7]
            keys[g, s % 2, : len(a)] = a
            lens[g, s % 2] = len(a)
        eng.make_actions(keys, lens)
        ref.make_actions(keys, lens)
        d1 = eng.finish_actions(10)
        d2 = ref.finish_actions(10)
        rec = eng.observe()[0]
        if flagged_at is None and rec["fifo_overflow"][:half, 1].any():
            flagged_at = s
            assert rec["fifo_overflow"][:half, 1].all() and rec["fifo_len"][:half, 1].min() == 8      # same scenario in every flooded game
            assert d1[:half].all() and not d2[:half].any()                                             # their round is over; the reference's goes on
            assert not rec["fifo_overflow"][half:].any() and not rec["fifo_overflow"][:half, 0].any()
            assert eng.take_errors() == 1 and eng.take_errors() == 0                                   # TETRIS_ERR_FIFO, reported once
        if flagged_at is None:
            assert np.array_equal(d1, d2), s
            engines.assert_same_state(eng, ref, where=f"step {s}")
        else:
            assert np.array_equal(d1[half:], d2[half:]), s
            engines.assert_same_state(eng, ref, idx=np.arange(half, n, dtype=np.int32), where=f"untouched games, step {s}")
        idx = np.nonzero(d2)[0].astype(np.int32)                 # ordinary game-overs (the oracle's): reset on both sides
        idx = idx[idx >= half] if flagged_at is not None else idx
        if len(idx):
            eng.reset(idx, 500 + s)
            ref.reset(idx, 500 + s)
    assert flagged_at is not None and 35 < flagged_at < 60
    assert int(ref.observe()[0]["fifo_len"][:half, 1].max()) > 8                                       # the reference's queue did grow past 8
    # a reset clears the error: the flagged games play on, equal to the oracle again
    idx = np.arange(half, dtype=np.int32)
    eng.reset(idx, 77)
    ref.reset(idx, 77)
    for s in range(20):
        rot, trans = rng.integers(0, 4, n).astype(np.uint8), rng.integers(0, 10, n).astype(np.uint8)
        d1, d2 = eng.step_rt(rot, trans, s % 2), ref.step_rt(rot, trans, s % 2)
        assert np.array_equal(d1, d2)
        idx = np.nonzero(d2)[0].astype(np.int32)
        if len(idx):
            eng.reset(idx, 900 + s)
            ref.reset(idx, 900 + s)
    engines.assert_same_state(eng, ref, where="after the reset of the flagged games")
    assert eng.take_errors() == 0
