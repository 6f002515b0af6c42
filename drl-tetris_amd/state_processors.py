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
