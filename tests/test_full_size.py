"""Parity at BASELINE.json's full sizes (GPU only): 65 536 games, one launch per env-step exactly as
bench.py runs them, against the oracle on all host cores — counters and the complete final state of
every board; plus size-independent properties of the batched path."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests import engines

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,steps", [(1, 96), (2, 96)])
def test_64k_boards_rollout_bit_exact(P, steps):
    n = 65536
    seeds = orc.episode_seed(np.arange(n), 0)
    eng = engines.make("hip", n, P, seeds=seeds)
    ref = engines.make("oracle", n, P, seeds=seeds)
    c_gpu, _ = eng.rollout_random(steps, 1)
    _, c_ref = ref.rollout_random(steps, threads=min(32, len(os.sched_getaffinity(0))))
    assert c_gpu.tolist() == c_ref.tolist()
    for lo in range(0, n, 8192):
        idx = np.arange(lo, lo + 8192, dtype=np.int32)
        engines.assert_same_state(eng, ref, idx=idx, where=f"games {lo}..")


@pytest.mark.parametrize("P,steps,fused", [(1, 3000, 30), (2, 2000, 20)])
def test_64k_boards_long_fused_rollout_bit_exact(P, steps, fused):
    """A soak at full size: 2-3 thousand env-steps per board (1.3-2.0e8 env-steps, ~1e7 episodes), `fused` steps per launch,
    against the oracle on all host cores — counters and every board's complete final state."""
    n = 65536
    seeds = orc.episode_seed(np.arange(n), 0)
    eng = engines.make("hip", n, P, seeds=seeds)
    ref = engines.make("oracle", n, P, seeds=seeds)
    c_gpu, _ = eng.rollout_random(steps // fused, fused)
    _, c_ref = ref.rollout_random(steps, threads=min(32, len(os.sched_getaffinity(0))))
    assert c_gpu.tolist() == c_ref.tolist() and int(c_gpu[0]) == n * steps
    for lo in range(0, n, 8192):
        idx = np.arange(lo, lo + 8192, dtype=np.int32)
        engines.assert_same_state(eng, ref, idx=idx, where=f"games {lo}..")


def test_batch_independence_and_order_invariance():
    """Games never interact: stepping a permuted / split batch gives the same per-game results
    (the property that makes sharding across GPUs collective-free)."""
    n, P = 8192, 2
    seeds = orc.episode_seed(np.arange(n), 0)
    a = engines.make("hip", n, P, seeds=seeds)
    perm = np.random.default_rng(0).permutation(n)
    b = engines.make("hip", n, P, seeds=seeds[perm])
    rng = np.random.default_rng(1)
    for s in range(64):
        rot, trans = rng.integers(0, 4, n).astype(np.uint8), rng.integers(0, 10, n).astype(np.uint8)
        da = a.step_rt(rot, trans, s % 2)
        db = b.step_rt(rot[perm], trans[perm], s % 2)
        assert np.array_equal(da[perm], db)
    ra, rb = a.observe()[0], b.observe()[0]
    for f in ra.dtype.names:                      # field-wise: struct padding is indeterminate
        assert np.array_equal(ra[f][perm], rb[f]), f
    # snapshot -> restore is the identity on the raw state words
    blob = a.snapshot()
    a.restore(blob)
    assert a.snapshot().tobytes() == blob.tobytes()
