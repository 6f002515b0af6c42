"""The give-up path of chained rollout launches (include/tetris_hip.h: tetris_set_chained).

A wave of a chained launch that cannot get its predecessor's epoch within its spin bound leaves its games untouched; the
call then finishes those games with the un-chained kernel and returns OK with the same results as ever, reports
TETRIS_ERR_CHAIN_FELL_BACK once and switches chaining off for the batch.  The worker of the reference shares its GPU with
the agent's network (drl_tetris/worker.py:91-118), so a predecessor that is late because something else holds the device
is an ordinary event, not an error."""
import os

import numpy as np
import pytest

from oracle import oracle as orc
from tests import engines

pytestmark = pytest.mark.gpu

FELL_BACK = 4


def _check(eng, ref, n, total, steps):
    _, want = ref.rollout_random(steps, threads=min(32, len(os.sched_getaffinity(0))))
    assert total.tolist() == want.tolist()
    for lo in range(0, n, 8192):
        idx = np.arange(lo, min(n, lo + 8192), dtype=np.int32)
        engines.assert_same_state(eng, ref, idx=idx, where=f"games {lo}..")


@pytest.mark.parametrize("P,n,direct", [(1, 65536, True), (2, 8192, True), (1, 65536, False), (2, 8192, False)])
def test_a_late_predecessor_makes_the_call_fall_back_and_the_results_stay_exact(P, n, direct):
    """Chain stream 1 is held by an idle kernel for 30 ms while the bound of a waiting wave is ~1 ms: the launches on the other
    streams give up, the call completes un-chained — OK, every board and the counters equal the oracle, fell-back bit reported."""
    seeds = orc.episode_seed(np.arange(n), 0)
    eng, ref = engines.make("hip", n, P, seeds=seeds), engines.make("oracle", n, P, seeds=seeds)
    assert eng.rollout_is_chained(1)
    eng.set_direct_dispatch(direct)                        # the library's own queues (calls of >= 16 launches) / the HIP streams
    total = np.zeros(4, np.uint64)
    c, _ = eng.rollout_random(40, 1)                       # chained, undisturbed
    total += c
    assert eng.take_errors() == 0 and eng.rollout_is_chained(1)
    eng.set_chain_spin_limit(2000)                         # ~1 ms
    before = eng.rollout_totals()                          # (synchronises: nothing may drain the streams between the stall and the launches)
    eng.debug_stall(1, 30000)                              # the second launch of the next call (and every third after it) starts 30 ms late
    stalled = 50 if P == 2 else 300                        # (300 on the streams: the call that is enqueued by one host thread per stream)
    eng.rollout_launch(stalled, 1, first_step=40)
    assert eng.rollout_was_direct() == direct
    total += eng.rollout_totals() - before
    assert eng.take_errors() == FELL_BACK
    assert eng.take_errors() == 0                          # reported once
    assert not eng.rollout_is_chained(1)                   # off until switched on again
    c, _ = eng.rollout_random(10, 1, first_step=40 + stalled)        # un-chained
    total += c
    eng.set_chained(True)
    eng.set_chain_spin_limit(0)
    assert eng.rollout_is_chained(1)
    c, _ = eng.rollout_random(28, 1, first_step=50 + stalled)        # chained again, from the epoch words the recovery left
    total += c
    assert eng.take_errors() == 0
    _check(eng, ref, n, total, 78 + stalled)


def test_a_co_tenant_that_holds_most_wave_slots_costs_time_not_results():
    """A kernel of `another tenant` holds 85 % of the device's wave slots for 20 ms while a chained call with a ~0.5 ms bound runs:
    whether or not waves give up (that depends on how the dispatcher deals the remaining slots), the call returns OK and every
    board and the counters equal the oracle."""
    n = 65536
    seeds = orc.episode_seed(np.arange(n), 0)
    eng, ref = engines.make("hip", n, 1, seeds=seeds), engines.make("oracle", n, 1, seeds=seeds)
    eng.set_chain_spin_limit(1000)
    total = np.zeros(4, np.uint64)
    step = 0
    for rep in range(3):
        eng.set_chained(True)
        before = eng.rollout_totals()
        eng.debug_stall(-1, 20000, 85)
        eng.rollout_launch(64, 1, first_step=step)
        total += eng.rollout_totals() - before
        step += 64
        assert eng.take_errors() in (0, FELL_BACK)
    _check(eng, ref, n, total, step)


def test_test_aid_and_knob_validate_their_arguments():
    pkg = __import__("__graft_entry__").package()
    eng = engines.make("hip", 256, 1, seeds=np.arange(256))
    for bad in ((4, 10, 0), (-2, 10, 0), (0, -1, 0), (0, 3000000, 0), (-1, 10, 101)):
        with pytest.raises(pkg.TetrisError):
            eng.debug_stall(*bad)
    eng.set_chain_spin_limit(123)
    eng.set_chain_spin_limit(0)                 # back to the default
    eng.debug_stall(3, 100)                     # the batch's own stream: just a short idle kernel
    c, _ = eng.rollout_random(5, 1)
    assert int(c[0]) == 5 * 256 and eng.take_errors() == 0
