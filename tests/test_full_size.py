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


def test_c4_enumerate_16384_boards_at_step_12():
    """BASELINE config 4 at its full size: 16 384 single-player boards taken at step 12 of their first episode
    under the synthetic policy (auto-reset inside), all 4 x 10 (rotation, column) drop afterstates of every board — validity,
    landing row, rows a finalize would clear and the stamped board's columns — against the oracle's cell-based enumeration
    (TestField.cpp:64-125)."""
    n = 16384
    seeds = orc.episode_seed(np.arange(n), 0)
    eng = engines.make("hip", n, 1, seeds=seeds)
    ref = engines.make("oracle", n, 1, seeds=seeds)
    c_gpu, _ = eng.rollout_random(12, 1)
    _, c_ref = ref.rollout_random(12, threads=min(32, len(os.sched_getaffinity(0))))
    assert c_gpu.tolist() == c_ref.tolist()
    v1, y1, c1, a1 = eng.enumerate_drops()
    v2, y2, c2, a2 = ref.enumerate_drops()
    assert np.array_equal(v1, v2)
    ok = v2.astype(bool)
    assert np.array_equal(y1[ok], y2[ok]) and np.array_equal(c1[ok], c2[ok])          # rows with valid == 0 are unspecified
    for lo in range(0, n, 2048):                           # columns -> cells: bit y of column c = cell (y, c)
        sl = slice(lo, lo + 2048)
        cells = ((a1[sl][..., None, :] >> np.arange(20, dtype=np.uint32)[:, None]) & 1).astype(bool)     # [n,4,10,20,10]
        assert np.array_equal(cells[ok[sl]], a2[sl][ok[sl]] > 0)
    assert 0.4 < ok.mean() < 0.8 and c2[ok].max() >= 1
    # the device-resident form an agent consumes (no host copies): same numbers
    import ctypes as C

    import torch
    dev = torch.device("cuda", 0)
    d_valid = torch.zeros(n * 40, dtype=torch.uint8, device=dev)
    d_land = torch.zeros(n * 40, dtype=torch.int8, device=dev)
    d_clr = torch.zeros(n * 40, dtype=torch.uint8, device=dev)
    d_after = torch.zeros(n * 40 * 10, dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    torch.cuda.synchronize()           # the fills above run on torch's stream, the kernel below on the batch's own
    eng._check(eng.lib.tetris_enumerate_drops_dev(eng._h, None, n, None, p(d_valid), p(d_land), p(d_clr), p(d_after)))
    eng.sync()
    assert np.array_equal(d_valid.cpu().numpy().reshape(n, 4, 10), v2)
    assert np.array_equal(d_land.cpu().numpy().reshape(n, 4, 10)[ok], y2[ok])
    assert np.array_equal(d_clr.cpu().numpy().reshape(n, 4, 10)[ok], c2[ok])
    assert np.array_equal(d_after.cpu().numpy().view(np.uint32).reshape(n, 4, 10, 10)[ok], a1[ok])
    # ... and its planar form (rotation-minor column planes: [n][10][4] and [10][n][10][4])
    for t in (d_valid, d_land, d_clr, d_after):
        t.zero_()
    torch.cuda.synchronize()
    eng.enumerate_drops_dev(n, p(d_valid), p(d_land), p(d_clr), p(d_after), planar=True)
    eng.sync()
    assert np.array_equal(d_valid.cpu().numpy().reshape(n, 10, 4).transpose(0, 2, 1), v2)
    assert np.array_equal(d_land.cpu().numpy().reshape(n, 10, 4).transpose(0, 2, 1)[ok], y2[ok])
    assert np.array_equal(d_clr.cpu().numpy().reshape(n, 10, 4).transpose(0, 2, 1)[ok], c2[ok])
    assert np.array_equal(d_after.cpu().numpy().view(np.uint32).reshape(10, n, 10, 4).transpose(1, 3, 2, 0)[ok], a1[ok])


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


def test_chained_launches_bit_exact_and_only_where_the_launches_in_flight_fit():
    """Chained rollout launches (tetris_set_chained: three streams, per-wave epoch words) against the oracle, switching the mode
    back and forth in one batch; the library chains only where the launches in flight (three) fit on the device together."""
    n = 65536
    seeds = orc.episode_seed(np.arange(n), 0)
    eng = engines.make("hip", n, 1, seeds=seeds)
    ref = engines.make("oracle", n, 1, seeds=seeds)
    assert eng.rollout_is_chained(1) and eng.rollout_is_chained(8)
    total = np.zeros(4, np.uint64)
    step = 0
    for chained, launches, fused in ((True, 50, 1), (False, 30, 1), (True, 5, 8), (True, 40, 1), (False, 3, 8), (True, 31, 1)):
        eng.set_chained(chained)
        assert eng.rollout_is_chained(fused) == chained
        c, _ = eng.rollout_random(launches, fused, first_step=step)
        total += c
        step += launches * fused
    _, want = ref.rollout_random(step, threads=min(32, len(os.sched_getaffinity(0))))
    assert total.tolist() == want.tolist()
    for lo in range(0, n, 8192):
        engines.assert_same_state(eng, ref, idx=np.arange(lo, lo + 8192, dtype=np.int32), where=f"games {lo}..")
    # two-player boards (k_duo, 32 games per wave): 64k games are 2048 waves per launch; two such launches fit on the device
    # together (the chained kernel is held to 96 registers per lane: 5 waves per SIMD), so they chain over two streams — the
    # full-size rollout against the oracle is test_64k_boards_rollout_bit_exact[2-96]; fused launches never chain for two players
    big = engines.make("hip", n, 2, seeds=seeds)
    assert big.rollout_is_chained(1) and not big.rollout_is_chained(4)
    big.close()
    m = 8192
    small, sref = engines.make("hip", m, 2, seeds=seeds[:m]), engines.make("oracle", m, 2, seeds=seeds[:m])
    assert small.rollout_is_chained(1) and not small.rollout_is_chained(4)
    c1, _ = small.rollout_random(150, 1)
    small.set_chained(False)
    c2, _ = small.rollout_random(50, 1, first_step=150)
    _, want = sref.rollout_random(200, threads=8)
    assert (c1 + c2).tolist() == want.tolist()
    engines.assert_same_state(small, sref, where="two-player, chained then not")


@pytest.mark.parametrize("P", [1, 2])
def test_chained_launches_at_batch_sizes_that_are_not_whole_waves(P):
    """Chained launches (three streams, one epoch word per wave) on batches whose last wave is partly empty, and on batches
    smaller than a wave: counters and every board against the oracle, single steps and fused ones."""
    for n in (1, 63, 65, 1000, 5001):
        seeds = orc.episode_seed(np.arange(n), 0)
        eng, ref = engines.make("hip", n, P, seeds=seeds), engines.make("oracle", n, P, seeds=seeds)
        assert eng.rollout_is_chained(1)
        c1, _ = eng.rollout_random(150, 1)
        c2, _ = (eng.rollout_random(5, 10, first_step=150) if P == 1 else eng.rollout_random(50, 1, first_step=150))
        _, want = ref.rollout_random(200, threads=8)
        assert (c1 + c2).tolist() == want.tolist(), n
        engines.assert_same_state(eng, ref, where=f"n={n}")
        eng.close()


def test_chain_depth_two_where_three_launches_do_not_fit():
    """98 304 single-player boards = 1 536 waves per launch: three launches (4 608 waves) do not fit in the device's wave slots, two
    do — the library then rotates over two streams instead of three (or not chaining at all).  Counters and every board against the
    oracle."""
    n = 98304
    seeds = orc.episode_seed(np.arange(n), 0)
    eng, ref = engines.make("hip", n, 1, seeds=seeds), engines.make("oracle", n, 1, seeds=seeds)
    assert eng.rollout_is_chained(1)
    c1, _ = eng.rollout_random(70, 1)
    c2, _ = eng.rollout_random(3, 10, first_step=70)
    _, want = ref.rollout_random(100, threads=min(32, len(os.sched_getaffinity(0))))
    assert (c1 + c2).tolist() == want.tolist()
    for lo in range(0, n, 8192):
        engines.assert_same_state(eng, ref, idx=np.arange(lo, lo + 8192, dtype=np.int32), where=f"games {lo}..")


@pytest.mark.parametrize("P,direct", [(1, False), (2, False), (1, True)])
def test_long_chained_calls_stay_bit_exact_on_either_launch_path(P, direct):
    """Long calls on the stream path: from 256 launches per call on, tetris_rollout_launch enqueues every chain stream's launches
    from a thread of its own (a launch costs the host up to 4 us, the GPU needs one every 4: one thread cannot always keep up).
    By default such calls do not go through hipLaunchKernel at all but through the library's own queues (direct = True).  Two calls
    of 700 and 300 launches at 64k games, then a short one: counters and every board against the oracle."""
    n = 65536
    seeds = orc.episode_seed(np.arange(n), 0)
    eng, ref = engines.make("hip", n, P, seeds=seeds), engines.make("oracle", n, P, seeds=seeds)
    assert eng.rollout_is_chained(1)
    if not direct:
        eng.set_direct_dispatch(False)
    total = np.zeros(4, np.uint64)
    step = 0
    for launches in (700, 300, 40):
        c, _ = eng.rollout_random(launches, 1, first_step=step)
        assert eng.rollout_was_direct() == direct
        total += c
        step += launches
    assert eng.take_errors() == 0
    _, want = ref.rollout_random(step, threads=min(32, len(os.sched_getaffinity(0))))
    assert total.tolist() == want.tolist()
    for lo in range(0, n, 8192):
        engines.assert_same_state(eng, ref, idx=np.arange(lo, lo + 8192, dtype=np.int32), where=f"games {lo}..")
