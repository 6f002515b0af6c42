"""Replays a golden trace (tests/golden/trace_*.npz, produced by the compiled reference) through
any engine exposing reset / make_actions / finish_actions / observe, and compares every field
the reference can show, event by event.  Shared by the oracle tests (CPU) and the HIP tests (GPU)."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# what the compiled reference exposes (see tests/golden/ref_driver.py)
VISIBLE = ["field", "grid", "x", "y", "next", "dead", "reward", "inc_count", "combo_count", "combo_remaining"]
HIDDEN = ["spawn_rot", "cur_rot", "big", "tile", "piece", "lines_sent", "lines_cleared", "lines_blocked", "max_combo",
          "fifo_len", "fifo_count", "fifo_delay", "min_remaining", "drop_delay", "drop_time", "speedup_time",
          "lock_time", "lock_armed", "time_ms", "incoming", "lines_cleared_seen"]
COLOUR_ONLY = ["garbage_cleared"]      # needs cell colours (8 = garbage)


def trace_names():
    return sorted(os.path.basename(p)[len("trace_"):-len(".npz")] for p in glob.glob(os.path.join(GOLDEN, "trace_*.npz")))


def load_trace(name):
    return np.load(os.path.join(GOLDEN, f"trace_{name}.npz"))


def compare(got, want, fields, occupancy_only=False, where=""):
    for f in fields:
        a, b = got[f], want[f]
        if f == "field" and occupancy_only:
            a, b = a > 0, b > 0
        if not np.array_equal(a, b):
            raise AssertionError(f"{where}: field '{f}' differs\n got  {a.tolist() if a.size < 64 else a}\n want {b.tolist() if b.size < 64 else b}")


def replay(trace, make_engine, fields=None, occupancy_only=False, max_events=None, check_actions=False):
    """make_engine(n_players, height, width, pieces, seed) -> engine with one game."""
    P, H = int(trace["n_players"]), int(trace["height"])
    fields = fields or (VISIBLE + HIDDEN)
    kinds, seeds, players = trace["ev_kind"], trace["ev_seed"], trace["ev_player"]
    keys, lens, dones = trace["ev_keys"], trace["ev_len"], trace["ev_done"]
    want, want_ro, want_lw = trace["records"], trace["round_over"], trace["last_winner"]
    eng = None
    first_step_seen = False
    n = len(kinds) if max_events is None else min(len(kinds), max_events)
    has_actions = check_actions and "act_n" in trace.files
    if has_actions:      # NpzFile decompresses on every access: read once
        act_keys, act_lens, act_n, act_player = trace["act_keys"], trace["act_lens"], trace["act_n"], trace["act_player"]
    for e in range(n):
        k = int(kinds[e])
        if has_actions and k == 1:
            # get_actions(player) of the state BEFORE this event's action (recorded from the reference)
            want_lists = [act_keys[e][i, : act_lens[e][i]].tolist() for i in range(int(act_n[e]))]
            got_lists = eng.get_actions(0, int(act_player[e]))
            assert got_lists == want_lists, f"event {e}: get_actions differs ({len(got_lists)} vs {len(want_lists)} lists)"
        if k == 2:
            eng = make_engine(P, H, int(trace["width"]), trace["pieces"].tolist(), int(seeds[e]))
        elif k == 0:
            eng.reset(None, seeds=int(seeds[e]))
        else:
            K = np.zeros((1, P, keys.shape[1]), np.uint8)
            L = np.ones((1, P), np.uint8)        # the other players get [0] (tetris_environment.py:106-108)
            K[0, players[e], :] = keys[e]
            L[0, players[e]] = lens[e]
            eng.make_actions(K, L)
            done = eng.finish_actions(int(trace["ms"]))
            assert bool(done[0]) == bool(dones[e]), f"event {e}: done {done[0]} != {dones[e]}"
        rec, ro, lw = eng.observe()
        fs = fields
        if not first_step_seen:
            # ComboCounter::remaining is uninitialised until the first finish_action
            fs = [f for f in fields if f != "combo_remaining"]
            first_step_seen = k == 1
        compare(rec[0], want[e], fs, occupancy_only, where=f"event {e} (kind {k})")
        assert int(ro[0]) == int(want_ro[e]), f"event {e}: round_over"
        assert int(lw[0]) == int(want_lw[e]), f"event {e}: last_winner {lw[0]} != {want_lw[e]}"
    return n


def trace_groups():
    """Golden traces grouped by batch geometry (players, height, width, piece map, tick): the traces of one group are
    replayed as the games of ONE batch."""
    groups = {}
    for name in trace_names():
        t = load_trace(name)
        key = (int(t["n_players"]), int(t["height"]), int(t["width"]), tuple(t["pieces"].tolist()), int(t["ms"]))
        groups.setdefault(key, []).append(name)
    return groups


def replay_batch(names, make_engine, fields=None, occupancy_only=False, check_actions=False):
    """All traces `names` (same geometry) at FULL length as the games of one batch: per event index one reset call for the
    games whose event is a reset, one make_actions + finish_actions pair for the games whose event is an action, one
    observe — every field of every game compared with the reference's record after every event.
    make_engine(n_games, n_players, height, width, pieces, seeds) -> engine.  Returns the number of (game, event) pairs."""
    traces = [load_trace(n) for n in names]
    t0 = traces[0]
    P, H, W, ms = int(t0["n_players"]), int(t0["height"]), int(t0["width"]), int(t0["ms"])
    fields = fields or (VISIBLE + HIDDEN)
    # NpzFile decompresses on every access: read once
    T = [{k: t[k] for k in ("ev_kind", "ev_seed", "ev_player", "ev_keys", "ev_len", "ev_done", "records", "round_over", "last_winner")}
         for t in traces]
    A = [({k: t[k] for k in ("act_keys", "act_lens", "act_n", "act_player")} if check_actions and "act_n" in t.files else None) for t in traces]
    for t in T:
        assert int(t["ev_kind"][0]) == 2 and not (t["ev_kind"][1:] == 2).any()
    n = len(traces)
    eng = make_engine(n, P, H, W, t0["pieces"].tolist(), np.array([int(t["ev_seed"][0]) for t in T]))
    stepped = np.zeros(n, bool)                       # ComboCounter::remaining is uninitialised until the first finish_action
    K = max(t["ev_keys"].shape[1] for t in T)
    pairs = 0
    for e in range(max(len(t["ev_kind"]) for t in T)):
        live = [g for g in range(n) if e < len(T[g]["ev_kind"])]
        resets = [g for g in live if int(T[g]["ev_kind"][e]) == 0]
        acts = [g for g in live if int(T[g]["ev_kind"][e]) == 1]
        for g in acts:
            if A[g] is not None:
                a = A[g]
                want_lists = [a["act_keys"][e][i, : a["act_lens"][e][i]].tolist() for i in range(int(a["act_n"][e]))]
                got_lists = eng.get_actions(g, int(a["act_player"][e]))
                assert got_lists == want_lists, f"{names[g]} event {e}: get_actions differs ({len(got_lists)} vs {len(want_lists)} lists)"
        if resets:
            eng.reset(np.array(resets, np.int32), seeds=np.array([int(T[g]["ev_seed"][e]) for g in resets]))
        if acts:
            keys = np.zeros((len(acts), P, K), np.uint8)
            lens = np.ones((len(acts), P), np.uint8)       # the other players get [0] (tetris_environment.py:106-108)
            for j, g in enumerate(acts):
                pl = int(T[g]["ev_player"][e])
                keys[j, pl, : T[g]["ev_keys"].shape[1]] = T[g]["ev_keys"][e]
                lens[j, pl] = T[g]["ev_len"][e]
            idx = np.array(acts, np.int32)
            eng.make_actions(keys, lens, idx=idx)
            done = eng.finish_actions(ms, idx=idx)
            for j, g in enumerate(acts):
                assert bool(done[j]) == bool(T[g]["ev_done"][e]), f"{names[g]} event {e}: done {done[j]} != {T[g]['ev_done'][e]}"
        rec, ro, lw = eng.observe(np.array(live, np.int32))
        for j, g in enumerate(live):
            fs = fields if stepped[g] else [f for f in fields if f != "combo_remaining"]
            compare(rec[j], T[g]["records"][e], fs, occupancy_only, where=f"{names[g]} event {e}")
            assert int(ro[j]) == int(T[g]["round_over"][e]), f"{names[g]} event {e}: round_over"
            assert int(lw[j]) == int(T[g]["last_winner"][e]), f"{names[g]} event {e}: last_winner {lw[j]} != {T[g]['last_winner'][e]}"
            pairs += 1
        for g in acts:
            stepped[g] = True
    return pairs
