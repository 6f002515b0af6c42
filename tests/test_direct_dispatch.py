"""Direct dispatch of the chained rollout launches (include/tetris_hip.h: tetris_set_direct_dispatch; csrc/tetris_aql.h): the library
writes the launches' AQL packets into HSA queues of the batch's own instead of calling hipLaunchKernel on streams.  Same kernels,
same hand-over protocol — every board must come out as the oracle's (PythonHandle.cpp:149-188 per step), however calls through the
queues are mixed with work on the batch's HIP stream."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests import engines

pytestmark = pytest.mark.gpu

THREADS = min(32, len(os.sched_getaffinity(0)))


def _same(eng, ref, n, where=""):
    for lo in range(0, n, 8192):
        idx = np.arange(lo, min(n, lo + 8192), dtype=np.int32)
        engines.assert_same_state(eng, ref, idx=idx, where=f"{where} games {lo}..")


@pytest.mark.parametrize("P,n", [(1, 65536), (2, 65536), (1, 1000), (1, 70), (2, 33)])
def test_direct_dispatch_is_what_runs_and_is_bit_exact(P, n):
    """Chained calls go through the library's own queues (asserted, not assumed); 150 launches in calls of 1, 2, 3, 7, 20
    and 117 launches (fewer launches than queues; a call's first and last packet on the same queue) against the oracle."""
    seeds = orc.episode_seed(np.arange(n), 0)
    eng, ref = engines.make("hip", n, P, seeds=seeds), engines.make("oracle", n, P, seeds=seeds)
    assert eng.rollout_is_chained(1)
    eng.set_direct_dispatch(True, min_launches=1)          # (by default only calls of >= 16 launches go this way)
    total, step = np.zeros(4, np.uint64), 0
    for launches in (1, 2, 3, 7, 20, 117):
        c, ms = eng.rollout_random(launches, 1, first_step=step)
        assert eng.rollout_was_direct()
        assert eng.rollout_was_affine()                      # the XCD-affine kernels (tetris_set_xcd_affine)
        assert ms > 0.0
        total += c
        step += launches
    _, want = ref.rollout_random(step, threads=THREADS)
    assert total.tolist() == want.tolist()
    _same(eng, ref, n)
    assert eng.take_errors() == 0 and eng.rollout_is_chained(1)


@pytest.mark.parametrize("P", [1, 2])
def test_queues_and_streams_take_turns(P):
    """Calls through the queues alternate with calls through the streams and with reads on the batch's HIP stream, then a
    snapshot / restore round trip (the host writes the state between two direct calls): everything is ordered, nothing is read stale."""
    n = 16384
    seeds = orc.episode_seed(np.arange(n), 0)
    eng, ref = engines.make("hip", n, P, seeds=seeds), engines.make("oracle", n, P, seeds=seeds)
    total, step = np.zeros(4, np.uint64), 0
    for rep in range(6):
        eng.set_direct_dispatch(rep % 2 == 0, min_launches=1)
        c, _ = eng.rollout_random(13 + rep, 1, first_step=step)
        assert eng.rollout_was_direct() == (rep % 2 == 0)
        total += c
        step += 13 + rep
        eng.observe(np.arange(0, n, 7, dtype=np.int32))       # a kernel + copy on the batch's stream in between
    eng.set_direct_dispatch(True, min_launches=1)
    blob = eng.snapshot()
    eng.rollout_random(9, 1, first_step=step)              # moves on ...
    eng.restore(blob)                                      # ... and is put back by the host
    c, _ = eng.rollout_random(31, 1, first_step=step)
    assert eng.rollout_was_direct()
    total += c
    step += 31
    _, want = ref.rollout_random(step, threads=THREADS)
    assert total.tolist() == want.tolist()
    _same(eng, ref, n)


def test_a_long_call_wraps_the_argument_ring_and_the_flow_control():
    """5 000 launches in one call: 1 667 per queue against 512 argument slots and a run-ahead of ~80 packets per queue."""
    n = 4096
    seeds = orc.episode_seed(np.arange(n), 0)
    eng, ref = engines.make("hip", n, 1, seeds=seeds), engines.make("oracle", n, 1, seeds=seeds)
    c, _ = eng.rollout_random(5000, 1)
    assert eng.rollout_was_direct()                        # the default: calls of >= 16 launches
    _, want = ref.rollout_random(5000, threads=THREADS)
    assert c.tolist() == want.tolist()
    _same(eng, ref, n)


def test_same_results_with_direct_dispatch_off_and_short_calls_stay_on_the_streams():
    n = 8192
    seeds = orc.episode_seed(np.arange(n), 0)
    a, b = engines.make("hip", n, 2, seeds=seeds), engines.make("hip", n, 2, seeds=seeds)
    b.set_direct_dispatch(False)
    ca, _ = a.rollout_random(200, 1)
    cb, _ = b.rollout_random(200, 1)
    assert a.rollout_was_direct() and not b.rollout_was_direct()
    assert ca.tolist() == cb.tolist()
    assert np.array_equal(a.snapshot(), b.snapshot())
    a.rollout_random(8, 1, first_step=200)
    assert not a.rollout_was_direct()                      # below the default threshold


def test_xcd_affine_launches_equal_the_plain_hand_over_and_misplaced_launches_cost_time_not_results():
    """The XCD-affine chained kernel (include/tetris_hip.h: tetris_set_xcd_affine) against the write-through one on the same games;
    then the kernels are told start XCDs that are off by three: every workgroup finds itself misplaced and touches nothing, the call
    is finished un-chained — exact, and not the caller's business: no error bit, chaining stays on — and after three such calls
    the affine form is off for the batch."""
    n = 65536
    seeds = orc.episode_seed(np.arange(n), 0)
    a, b = engines.make("hip", n, 1, seeds=seeds), engines.make("hip", n, 1, seeds=seeds)
    ref = engines.make("oracle", n, 1, seeds=seeds)
    b.set_xcd_affine(False)
    total = np.zeros(4, np.uint64)
    ca, _ = a.rollout_random(400, 1)
    cb, _ = b.rollout_random(400, 1)
    assert a.rollout_was_affine() and b.rollout_was_direct() and not b.rollout_was_affine()
    assert ca.tolist() == cb.tolist()
    assert np.array_equal(a.snapshot(), b.snapshot())
    total += ca
    a.debug_xcd_skew(3)
    step = 400
    for rep in range(3):
        c, _ = a.rollout_random(20, 1, first_step=step)
        assert a.rollout_was_affine()                                  # tried, every workgroup stood aside, finished un-chained
        assert a.take_errors() == 0 and a.rollout_is_chained(1)
        total += c
        step += 20
    a.debug_xcd_skew(0)
    c, _ = a.rollout_random(40, 1, first_step=step)
    assert a.rollout_was_direct() and not a.rollout_was_affine()      # three strikes: the affine form is off
    total += c
    _, want = ref.rollout_random(500, threads=THREADS)
    assert total.tolist() == want.tolist()
    _same(a, ref, n)


def test_xcd_affine_state_survives_other_kernels_boundaries():
    """While an affine call keeps its games' state dirty in the XCDs' L2s (no cache maintenance between its own launches), another
    batch of the process steps, observes and copies on HIP streams from a second host thread — kernels whose starts invalidate and whose
    ends write back the caches.  The affine batch must come out as the oracle's: a kernel boundary of somebody else never costs it data."""
    import threading
    n = 65536
    seeds = orc.episode_seed(np.arange(n), 0)
    eng, ref = engines.make("hip", n, 1, seeds=seeds), engines.make("oracle", n, 1, seeds=seeds)
    other = engines.make("hip", 4096, 2, seeds=orc.episode_seed(np.arange(4096), 7))
    other.set_chained(False)
    stop = threading.Event()
    laps = [0]

    def noise():
        rng = np.random.default_rng(3)
        while not stop.is_set():
            other.step_rt(rng.integers(0, 4, 4096).astype(np.uint8), rng.integers(0, 10, 4096).astype(np.uint8), np.zeros(4096, np.uint8))
            other.observe(np.arange(0, 4096, 5, dtype=np.int32))
            other.snapshot(np.arange(64, dtype=np.int32))
            laps[0] += 1

    t = threading.Thread(target=noise)
    t.start()
    try:
        total, step = np.zeros(4, np.uint64), 0
        for launches in (3000, 20, 2000):
            c, _ = eng.rollout_random(launches, 1, first_step=step)
            assert eng.rollout_was_affine()
            total += c
            step += launches
    finally:
        stop.set()
        t.join()
    assert laps[0] > 20                                    # the other batch really ran meanwhile
    assert eng.take_errors() == 0
    _, want = ref.rollout_random(step, threads=THREADS)
    assert total.tolist() == want.tolist()
    _same(eng, ref, n)
