"""Engine factories for the parity tests.

  "hip"      the product: drl-tetris_amd/lib/libtetris_hip.so on a real MI355X (tests marked gpu)
  "harness"  tests/cpu_harness: the product's kernel bodies compiled by g++ (CPU suite; test-only)
  "oracle"   oracle/liboracle.so, the checker
"""
import pytest

import __graft_entry__ as ge
from oracle import oracle as orc

ENGINE_PARAMS = ["harness", pytest.param("hip", marks=pytest.mark.gpu)]


def make(kind, n_games, n_players, height=20, pieces=(0, 1, 2, 3, 4, 5, 6), seeds=None, colours=False):
    if kind == "oracle":
        return orc.OracleBatch(n_games, n_players, height, 10, pieces=pieces, seeds=seeds)
    pkg = ge.package()
    if kind == "harness":
        return pkg.TetrisBatch(n_games, n_players, height, 10, pieces=pieces, seeds=seeds, lib_path=ge.build_harness(), colours=colours)
    if kind == "hip":
        return pkg.TetrisBatch(n_games, n_players, height, 10, pieces=pieces, seeds=seeds, device=0, colours=colours)
    raise ValueError(kind)


# record fields the bitboard engine tracks (no cell colours => no garbage_cleared, field as occupancy;
# weights are folded into the RNG tables)
ENGINE_FIELDS = ["field", "grid", "x", "y", "piece", "tile", "spawn_rot", "cur_rot", "big", "next", "dead", "reward",
                 "inc_count", "combo_count", "combo_remaining", "lock_armed", "fifo_len", "line_count", "time_ms",
                 "incoming", "drop_delay", "drop_time", "speedup_time", "lock_time", "min_remaining", "combo_start",
                 "combo_time", "fifo_delay", "fifo_count", "lines_sent", "lines_cleared", "lines_blocked", "max_combo",
                 "lines_cleared_seen", "piece_draws", "hole_draws"]


def assert_same_state(eng, ref, idx=None, where=""):
    import numpy as np
    a, ro_a, lw_a = eng.observe(idx)
    b, ro_b, lw_b = ref.observe(idx)
    for f in ENGINE_FIELDS:
        fa, fb = (a[f] > 0, b[f] > 0) if f == "field" else (a[f], b[f])
        if not np.array_equal(fa, fb):
            bad = np.argwhere(np.asarray(fa != fb).reshape(fa.shape[0], fa.shape[1], -1).any(axis=2))
            g, p = bad[0]
            raise AssertionError(f"{where}: '{f}' differs for {len(bad)} boards; first game {g} player {p}:\n got  {a[f][g, p]}\n want {b[f][g, p]}")
    assert np.array_equal(ro_a, ro_b), f"{where}: round_over"
    assert np.array_equal(lw_a, lw_b), f"{where}: last_winner"


PATH_COUNTERS = ["kick", "kick_2nd", "kick_3rd", "kick_failed", "kick_down", "drop_exact", "rt_off_spawn",
                 "garbage_row", "garbage_lift2", "death_garbage", "death_spawn", "timer_lock", "key_kick", "key_kick_failed",
                 "undo_simple", "undo_full", "undo_mispredict"]


def harness_path_counts():
    """Key-interpreter path counters of the CPU harness build since the last call (test-only; tetris_engine.h TE_COUNT).
    Not thread-safe: meaningful for single-threaded harness calls, which is how the harness runs."""
    import ctypes as C
    lib = C.CDLL(ge.build_harness())
    out = (C.c_ulonglong * len(PATH_COUNTERS))()
    lib.harness_path_counts(out)
    return dict(zip(PATH_COUNTERS, [int(v) for v in out]))
