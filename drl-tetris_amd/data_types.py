"""Host-side value types of the drop-in API: action, action_list, rewards and state.

Same names and behaviour as the reference's environment/data_types/*.py (action.py:2-9,
action_list.py:3-37, reward.py:7-77, state.py:1-40) so that agent code written against the
reference keeps working; the implementation is new and the `state` holds a GPU snapshot blob plus
its decoded record instead of a deep copy of a C++ PythonHandle.
"""
import numpy as np

from . import layout


class action(list):
    """A key sequence (ints 0..10, PythonHandle.cpp:73-112).  Reference: data_types/action.py:2-7."""

    def __init__(self, container):
        assert type(container) in (list, action), f"action created from non-list type object...: x={container}, type(x)={type(container)}"
        super().__init__(container)

    def __str__(self):
        return "action(" + super().__str__() + ")"


null_action = action([0])


class action_list:
    """Ordered set of actions with the reference's null-move policy (data_types/action_list.py:3-37):
    duplicates are dropped keeping first occurrences; unless the input already contains the null action it is put in
    front; `remove_null` then strips every null action again as long as something else remains."""

    def __init__(self, container=None, remove_null=False):
        self.remove_null = remove_null
        source = list(container) if container is not None and len(container) > 0 else [null_action]
        for item in source:
            assert type(item) in (action, list), (
                f"attempted to create action list with non-list type actions (x={item} type(x)={type(item)}).")
        ordered = ([] if null_action in source else [null_action]) + [action(item) for item in source]
        self.container = []
        for item in ordered:
            if item not in self.container:
                self.container.append(item)
        if remove_null:
            self.remove_nulls()

    def remove_nulls(self):
        others = [a for a in self.container if a != null_action]
        if others:
            self.container = others
        else:
            self.container = self.container[:1]

    def __add__(self, other):
        return action_list(other.container + self.container, remove_null=(other.remove_null and self.remove_null))

    def __getitem__(self, key):
        return null_action if key > len(self.container) else self.container[key]       # sic: action_list.py:27-32

    def __len__(self):
        return len(self.container)

    def __iter__(self):
        return iter(self.container)

    def __contains__(self, x):
        return x in self.container

    def __repr__(self):
        return f"action_list({self.container})"

    __str__ = __repr__


def _vec(x):
    return np.array(x).ravel()


class reward:
    """Vector-valued reward split into an extrinsic and an intrinsic part; behaves like the reference's
    data_types/reward.py:7-52: `r()` is the sum of both parts, `+`/`-` combine part-wise through the subclass rules,
    the positional form is reward(extrinsic) (a second positional value is accepted and ignored there too)."""

    def __init__(self, *args, **kwargs):
        if len(args) > 2:
            raise ValueError("reward needs to have 0, 1 or 2 arguments: [<extrinsic>] [<intrinsic>]")
        if args and "extrinsic" in kwargs:
            raise ValueError("Doubly specified extrinsic :(")
        self._extrinsic = _vec(args[0] if args else kwargs.pop("extrinsic"))
        self._intrinsic = _vec(kwargs.pop("intrinsic", np.zeros((1,))))

    extrinsic = property(lambda self: self._extrinsic)
    intrinsic = property(lambda self: self._intrinsic)
    reward = property(lambda self: self._extrinsic)

    def ext_rule(self, *args, **kwargs):
        raise ValueError("Dont use base-class!")

    def int_rule(self, *args, **kwargs):
        raise ValueError("Dont use base-class!")

    def _combine(self, other, **mode):
        return type(self)(extrinsic=self.ext_rule(self._extrinsic, other._extrinsic, **mode),
                          intrinsic=self.int_rule(self._intrinsic, other._intrinsic, **mode))

    def __add__(self, other):
        return self._combine(other, add=True)

    def __sub__(self, other):
        return self._combine(other, sub=True)

    def __call__(self, separate_components=False):
        if separate_components:
            return (self._extrinsic, self._extrinsic)       # sic: reward.py:42 returns the extrinsic part twice
        return self._extrinsic.sum() + self._intrinsic.sum()

    def __repr__(self):
        return f"reward<R={self()}=({self._extrinsic.tolist()}, {self._intrinsic.tolist()} )>"

    __str__ = __repr__


class standard_reward(reward):
    """Plain element-wise combination of both parts (reward.py:55-62)."""

    @staticmethod
    def _elementwise(x, y, add=False, sub=False):
        return x + y if add else x - y

    def ext_rule(self, x, y, **mode):
        return self._elementwise(x, y, **mode)

    def int_rule(self, x, y, **mode):
        return self._elementwise(x, y, **mode)


class maingoal_reward(standard_reward):
    """Of the other operand's extrinsic part only component 0 (the win/lose signal) takes part (reward.py:64-71)."""

    def ext_rule(self, x, y, **mode):
        main_only = np.zeros_like(y)
        main_only[0] = y[0]
        return self._elementwise(x, main_only, **mode)


class coopintrinsic_reward(maingoal_reward):
    def int_rule(self, x, y, **kwargs):
        return x + y


class _player_view:
    """What the reference's pybind `State` exposes for one player (PythonHandle.h:54-82), as numpy
    arrays decoded from a tetris_record."""

    __slots__ = ("field", "piece", "x", "y", "inc_lines", "combo_time", "combo_count", "nextpiece", "reward", "dead")

    def __init__(self, rec, height, width):
        self.field = rec["field"][:height, :width].copy()
        self.piece = rec["grid"].copy()
        self.x = np.array([rec["x"]], np.int8)
        self.y = np.array([rec["y"]], np.int8)
        self.inc_lines = np.array([rec["inc_count"]], np.uint8)
        self.combo_time = np.array([rec["combo_remaining"]], np.uint16)
        self.combo_count = np.array([rec["combo_count"]], np.uint8)
        self.nextpiece = np.array([rec["next"]], np.uint8)
        self.reward = np.array([rec["reward"]], np.uint8)
        self.dead = np.array([rec["dead"]], np.uint8)


class backend_snapshot:
    """Stand-in for the copied PythonHandle inside a reference `state`: `.states[p]` views plus the raw
    snapshot words that restore the exact game (RNG position included) on the GPU."""

    def __init__(self, blob, records, height, width, round_over=0, last_winner=-1):
        self.blob = np.array(blob, dtype=np.uint32, copy=True)
        self.records = records
        self.states = [_player_view(records[p], height, width) for p in range(len(records))]
        self.round_over = int(round_over)
        self.last_winner = int(last_winner)
        self.height, self.width = height, width

    def copy(self):
        b = backend_snapshot(self.blob, self.records.copy(), self.height, self.width, self.round_over, self.last_winner)
        for mine, theirs in zip(b.states, self.states):
            mine.dead[0] = theirs.dead[0]
        return b

    def sync_dead_to_blob(self):
        """Python may write `states[p].dead[0]` (state.py:11,16); carry it into the restore words."""
        # words per player-board come from the blob itself: 39, or 69 for batches with colour planes (tetris_layout.h)
        nw = (len(self.blob) - layout.NGWORDS) // len(self.states)
        for p, s in enumerate(self.states):
            w = layout.NGWORDS + p * nw + layout.W_PIECE
            self.blob[w] = (self.blob[w] & ~np.uint32(1 << 17)) | (np.uint32(1 << 17) if s.dead[0] else np.uint32(0))


class state:
    """A snapshot of one game plus a lazy per-player processor — same surface as the reference's data_types/state.py:1-40
    (`backend_state`, `is_dead`, `lock`/`unlock`, indexing by player, iteration over the raw per-player views)."""

    def __init__(self, backend_state, state_processor, unlocked=True):
        self.state_processor = state_processor
        self.backend_state = backend_state.copy()
        self.is_dead = [view.dead[0] for view in backend_state.states]
        self.unlocked = True
        if not unlocked:
            self.lock()

    def _write_dead(self, values):
        for view, value in zip(self.backend_state.states, values):
            view.dead[0] = value

    def lock(self):
        """Marks every player dead in the snapshot, so that actions performed on it change nothing (state.py:9-12)."""
        self._write_dead([1] * len(self))
        self.unlocked = False

    def unlock(self):
        self._write_dead(self.is_dead)
        self.unlocked = True

    def _players(self, idx):
        if isinstance(idx, slice):
            return list(range(*idx.indices(len(self))))
        return list(idx) if hasattr(idx, "__iter__") else None

    def __getitem__(self, idx):
        players = self._players(idx)
        if players is None:
            return self.state_processor(self.backend_state, idx)
        return [self.state_processor(self.backend_state, p) for p in players]

    def __len__(self):
        return len(self.backend_state.states)

    def __iter__(self):
        return iter(self.backend_state.states)
