"""state_processor: backend state -> what an agent sees.  Reference:
environment/env_utils/state_processors.py:23-85 (`state_dict`, `raw`, registry, wrapper class)."""
import numpy as np

# piece-grid value -> piece index ("the engine codes pieces differently in different places",
# state_processors.py:24; values from gamePlay.cpp:125-139)
GRID_VALUE_TO_PIECE = {1: 5, 2: 4, 3: 1, 4: 0, 5: 2, 6: 6, 7: 3}
MIRROR_GRID_VALUE_TO_PIECE = {4: 1, 3: 0, 5: 3, 7: 2, 2: 4, 1: 5, 6: 6}
MIRROR_PIECE = [1, 0, 3, 2, 4, 5, 6]


def state_dict(x, player, *parameters):
    _pieces, old_state_dict, separate_piece, augment = parameters[0]
    piece_set = list(range(8 if old_state_dict else 7))
    s = x.states[player]
    current = GRID_VALUE_TO_PIECE[int(s.piece.max())]
    ret = {
        "field": (np.array(s.field) > 0).astype(np.uint8),
        "piece": np.array([int(p == current) for p in piece_set]).astype(np.uint8),
        "x": np.array(s.x.copy(), dtype=np.uint8),
        "y": np.array(s.y.copy(), dtype=np.uint8),
        "incoming_lines": np.array(s.inc_lines),
        "combo_time": np.array(min(25000, s.combo_time + 50) // 100, dtype=np.uint8),
        "combo_count": np.array(s.combo_count, dtype=np.uint8),
        "nextpiece": np.array([int(p == s.nextpiece[0]) for p in piece_set], dtype=np.uint8),
    }
    if separate_piece:
        ret["piece_idx"] = current
    if augment:
        mirrored = MIRROR_GRID_VALUE_TO_PIECE[int(s.piece.max())]
        ret["aug"] = {
            "field": ret["field"][:, ::-1],
            "piece": np.array([int(p == mirrored) for p in piece_set]).astype(np.uint8),
            "nextpiece": np.array([MIRROR_PIECE[int(p == s.nextpiece[0])] for p in piece_set], dtype=np.uint8),   # sic: state_processors.py:50
            "piece_idx": mirrored,
        }
    return ret


class state_dict_batch:
    """`state_dict` for ALL games and players of a data_types.snapshot_batch at once (numpy over [n, P, ...]); `item(j, p)` hands
    out the dict `state_dict(backend_state_j, p, parameters)` would build — same keys, values, dtypes and shapes (the golden
    fixtures of the reference's own Python check them) — as views into the batch arrays: ~3 us per dict instead of ~23."""

    def __init__(self, batch, parameters):
        _pieces, old_state_dict, self.separate_piece, self.augment = parameters
        v = batch.views()
        k = 8 if old_state_dict else 7
        eye = np.eye(8, k, dtype=np.uint8)                       # row i = [int(p == i) for p in range(k)]
        kind = (batch.blob[:, 5:].reshape(len(batch), batch.n_players, batch.nw)[:, :, 10] & 7).astype(np.int64)     # W_PIECE bits 0..2
        self.kind = kind                                         # = GRID_VALUE_TO_PIECE[max of the piece grid] for kinds 0..6
        self.field = (v["field"] > 0).astype(np.uint8)
        self.piece = eye[np.minimum(kind, 7)]
        self.x = v["x"].astype(np.uint8)[..., None]
        self.y = v["y"].astype(np.uint8)[..., None]
        self.inc = v["inc_lines"][..., None]
        t = (v["combo_time"] + np.uint16(50))                    # uint16 array + 50 wraps like the reference's (state_processors.py:38)
        self.capped = t > 25000                                  # min(25000, array) then returns the Python int: a 0-d result
        self.combo_time = (np.minimum(t, 25000) // 100).astype(np.uint8)[..., None]
        self.combo_count = v["combo_count"][..., None]
        self.next = eye[np.minimum(v["nextpiece"].astype(np.int64), 7)]
        if self.augment:
            mirror = np.array(MIRROR_PIECE + [7], np.int64)
            self.aug_field = self.field[:, :, :, ::-1]
            self.aug_kind = mirror[np.minimum(kind, 7)]
            self.aug_piece = eye[self.aug_kind]
            # sic (state_processors.py:50): MIRROR_PIECE[int(p == next)] per p -> 1 everywhere except a 0 at p == next
            self.aug_next = np.where(self.next > 0, np.uint8(MIRROR_PIECE[1]), np.uint8(MIRROR_PIECE[0]))

    def item(self, j, p):
        if self.kind[j, p] > 6:
            raise KeyError(0)                                    # no piece yet: the reference's lookup of max(piece grid) = 0 fails too
        ret = {
            "field": self.field[j, p], "piece": self.piece[j, p], "x": self.x[j, p], "y": self.y[j, p],
            "incoming_lines": self.inc[j, p],
            "combo_time": np.array(250, dtype=np.uint8) if self.capped[j, p] else self.combo_time[j, p],
            "combo_count": self.combo_count[j, p], "nextpiece": self.next[j, p],
        }
        if self.separate_piece:
            ret["piece_idx"] = int(self.kind[j, p])
        if self.augment:
            ret["aug"] = {"field": self.aug_field[j, p], "piece": self.aug_piece[j, p], "nextpiece": self.aug_next[j, p],
                          "piece_idx": int(self.aug_kind[j, p])}
        return ret


def raw(x, player, *parameters):
    return x.states[player]


func_dict = {
    "raw": (raw, []),
    "state_dict": (state_dict, ["pieces", "old_state_dict", "state_processor_separate_piece", "augment_data"]),
}


class state_processor:
    def __init__(self, func, parameters=()):
        self.func = func
        self.parameters = list(parameters)

    def __call__(self, x, player):
        return self.func(x, player, self.parameters)
