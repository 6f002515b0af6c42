"""Drives the COMPILED REFERENCE (oracle/_ref, container-only build of /root/reference's C++
backend) and flattens everything it can show into oracle.RECORD arrays.

Used by make_golden.py (fixture generation, runs only where /root/reference exists) and by
tests that compare oracle <-> reference live when oracle/_ref was shipped prebuilt.
Nothing here reads /root/reference at run time; it only loads the built extension.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402


class RefGame:
    """One reference PythonHandle (PythonHandle.h:84-111) with an explicit seed."""

    def __init__(self, n_players=2, height=20, width=10, pieces=(0, 1, 2, 3, 4, 5, 6), seed=1000):
        self.mod, self.set_time = orc.ref_module()
        self.P, self.H, self.W = n_players, height, width
        self.pieces = (list(pieces) * 7)[:7]
        self.mod.set_pieces(self.pieces)          # tetris_environment.py:34-35
        self.set_time(int(seed))
        self.h = self.mod.PythonHandle(n_players, [height, width])

    def reset(self, seed):
        self.mod.set_pieces(self.pieces)
        self.set_time(int(seed))
        self.h.reset()

    def make(self, keys_per_player):
        self.h.make_action([list(map(int, k)) for k in keys_per_player])

    def finish(self, ms=400):
        return bool(self.h.finish_action(ms))

    def step(self, keys, player, ms=400):
        """tetris_environment.perform_action (tetris_environment.py:102-116)."""
        a = [[0] for _ in range(self.P)]
        a[player] = list(map(int, keys))
        self.h.make_action(a)
        return bool(self.h.finish_action(ms))

    def get_actions(self, player):
        self.h.get_actions(player)
        return [list(a) for a in self.h.masks[player].action]

    def record(self, hidden=True):
        """-> (records[P], round_over, last_winner).  Private members the reference cannot
        show (comboStart/comboTime/lineCount, cogP, RNG position) stay zero."""
        rec = np.zeros(self.P, dtype=orc.RECORD)
        st = self.h.__getstate__() if hidden else None
        for p in range(self.P):
            s = self.h.states[p]
            r = rec[p]
            r["field"][: self.H, : self.W] = np.array(s.field)
            r["grid"] = np.array(s.piece)
            r["x"], r["y"] = s.x[0], s.y[0]
            r["next"], r["dead"], r["reward"] = s.nextpiece[0], s.dead[0], s.reward[0]
            r["inc_count"], r["combo_count"], r["combo_remaining"] = s.inc_lines[0], s.combo_count[0], s.combo_time[0]
            if hidden:
                g = st[0][p].__getstate__()
                field, _, _, _, data, garbage, combo, drop, _nextp, incoming, _inc_cnt, time_ms, seen, _rew, _dead = g
                piece = field.__getstate__()[1].__getstate__()
                r["spawn_rot"], r["cur_rot"], r["big"], r["tile"], r["piece"] = piece[1], piece[2], piece[5], piece[6], piece[7]
                d = data.__getstate__()
                r["lines_sent"], r["garbage_cleared"], r["lines_cleared"], r["lines_blocked"], r["max_combo"] = d[0], d[3], d[4], d[6], d[7]
                q, min_rem, _add = garbage.__getstate__()
                r["fifo_len"] = len(q)
                for i, e in enumerate(q[: orc.FIFO_CAP]):
                    c, dl = e.__getstate__()
                    r["fifo_count"][i], r["fifo_delay"][i] = c, dl
                r["min_remaining"] = min_rem
                dd = drop.__getstate__()
                r["drop_delay"], r["drop_time"], r["speedup_time"], r["lock_time"], r["lock_armed"] = dd[0], dd[1], dd[3], dd[4], int(dd[5])
                r["time_ms"], r["incoming"], r["lines_cleared_seen"] = time_ms, incoming, seen
        ro = int(st[1]) if hidden else 0
        return rec, ro, int(self.h.last_winner)


# fields both the reference and the oracle can show (see RefGame.record)
VISIBLE = ["field", "grid", "x", "y", "next", "dead", "reward", "inc_count", "combo_count", "combo_remaining"]
HIDDEN = ["spawn_rot", "cur_rot", "big", "tile", "piece", "lines_sent", "garbage_cleared", "lines_cleared",
          "lines_blocked", "max_combo", "fifo_len", "fifo_count", "fifo_delay", "min_remaining", "drop_delay",
          "drop_time", "speedup_time", "lock_time", "lock_armed", "time_ms", "incoming", "lines_cleared_seen"]


def diff_records(a, b, fields, skip_combo_remaining=False):
    """Names of the fields in which two record arrays differ."""
    bad = []
    for f in fields:
        if skip_combo_remaining and f == "combo_remaining":
            continue
        if not np.array_equal(a[f], b[f]):
            bad.append(f)
    return bad
