"""Action sources for fixture generation and parity tests (test infrastructure).

`greedy_rt` plays a one-piece look-ahead height/holes heuristic on a scratch OracleBatch so
that traces contain line clears, combos, garbage traffic and long survival — regimes a
random policy never reaches (SURVEY.md §8c "regimes the fixtures must reach").
"""
import numpy as np

from oracle import oracle as orc


def rt_keys(r, t):
    """sventon_utils.py:9-13 make_action"""
    return [8] * int(r) + [2] + [3] * int(t) + [7]


class GreedyRT:
    def __init__(self, n_players, height, width=10, pieces=(0, 1, 2, 3, 4, 5, 6), sloppiness=0.0, seed=0):
        self.P, self.H = n_players, height
        self.scratch = orc.OracleBatch(40, n_players, height, width, pieces=pieces)
        self.rng = np.random.default_rng(seed)
        self.sloppiness = sloppiness
        self.rots = np.repeat(np.arange(4), 10).astype(np.uint8)
        self.trans = np.tile(np.arange(10), 4).astype(np.uint8)

    def choose(self, batch, game, player):
        if self.rng.random() < self.sloppiness:
            return int(self.rng.integers(4)), int(self.rng.integers(10))
        self.scratch.copy_from(batch, src_idx=np.full(40, game, np.int32))
        keys = np.zeros((40, self.P, 16), np.uint8)
        lens = np.zeros((40, self.P), np.uint8)
        for i in range(40):
            k = rt_keys(self.rots[i], self.trans[i])
            keys[i, player, : len(k)] = k
            lens[i, player] = len(k)
        self.scratch.make_actions(keys, lens)
        rec, _, _ = self.scratch.observe()
        f = rec["field"][:, player, : self.H, :] > 0            # [40, H, W]
        full = f.all(axis=2).sum(axis=1)
        heights = np.where(f.any(axis=1), self.H - f.argmax(axis=1), 0)   # [40, W]
        filled_below = np.cumsum(f, axis=1) > 0
        holes = (filled_below & ~f).sum(axis=(1, 2))
        bump = np.abs(np.diff(heights, axis=1)).sum(axis=1)
        score = -0.51 * heights.sum(axis=1) + 0.76 * full * 10 - 0.8 * holes * 4 - 0.18 * bump
        score = score + self.rng.random(40) * 1e-3
        best = int(np.argmax(score))
        return int(self.rots[best]), int(self.trans[best])
