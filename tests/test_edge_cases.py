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
