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
    """Ordered, de-duplicated list of actions with the reference's null-move policy
    (data_types/action_list.py:3-37): the null action is put first unless already present; with
    remove_null it is dropped again while other actions remain."""

    def __init__(self, container=None, remove_null=False):
        self.remove_null = remove_null
        if container is None or len(container) == 0:
            container = [null_action]
        self.container = [] if null_action in container else [null_action]
        for x in container:
            assert type(x) in (action, list), f"attempted to create action list with non-list type actions (x={x} type(x)={type(x)})."
            if x not in self.container:
                self.container.append(action(x))
        if self.remove_null:
            self.remove_nulls()

    def remove_nulls(self):
        while len(self.container) > 1 and null_action in self.container:
            self.container.remove(null_action)

    def __add__(self, n):
        return action_list(n.container + self.container, remove_null=(n.remove_null and self.remove_null))

    def __str__(self):
        return "action_list(" + str(self.container) + ")"

    __repr__ = __str__

    def __getitem__(self, key):
        if key > len(self.container):
            return null_action
        return self.container[key]

    def __len__(self):
        return len(self.container)

    def __iter__(self):
        return iter(self.container)

    def __contains__(self, x):
        return x in self.container


class reward:
    """Vector-valued reward with extrinsic/intrinsic parts (reference: data_types/reward.py:7-52)."""

    def __init__(self, *args, **kwargs):
        if len(args) > 2:
            raise ValueError("reward needs to have 0, 1 or 2 arguments: [<extrinsic>] [<intrinsic>]")
        if len(args) > 0:
            if "extrinsic" in kwargs:
                raise ValueError("Doubly specified extrinsic :(")
            self._extrinsic = np.array(args[0]).ravel()
        else:
            self._extrinsic = np.array(kwargs.pop("extrinsic")).ravel()
        # (the reference never reads a positional intrinsic: reward.py:14-21)
        self._intrinsic = np.array(kwargs.pop("intrinsic", np.zeros((1,)))).ravel()

    def ext_rule(self, *args, **kwargs):
        raise ValueError("Dont use base-class!")

    int_rule = ext_rule

    @property
    def reward(self):
        return self._extrinsic

    @property
    def extrinsic(self):
        return self._extrinsic

    @property
    def intrinsic(self):
        return self._intrinsic

    def __call__(self, separate_components=False):
        if separate_components:
            return (self._extrinsic, self._extrinsic)       # sic: reward.py:42
        return self._extrinsic.sum() + self._intrinsic.sum()

    def __add__(self, other):
        return type(self)(extrinsic=self.ext_rule(self._extrinsic, other._extrinsic, add=True),
                          intrinsic=self.int_rule(self._intrinsic, other._intrinsic, add=True))

    def __sub__(self, other):
        return type(self)(extrinsic=self.ext_rule(self._extrinsic, other._extrinsic, sub=True),
                          intrinsic=self.int_rule(self._intrinsic, other._intrinsic, sub=True))

    def __str__(self):
        return "reward<R=" + str(self()) + "=(" + str(self.extrinsic.tolist()) + ", " + str(self.intrinsic.tolist()) + " )>"

    __repr__ = __str__


class standard_reward(reward):
    def ext_rule(self, x, y, add=False, sub=False):
        return x + y if add else x - y

    int_rule = ext_rule


class maingoal_reward(standard_reward):
    """Only component 0 of the other operand's extrinsic part is combined (reward.py:64-71)."""

    def ext_rule(self, x, y, add=False, sub=False):
        tmp = np.zeros_like(y)
        tmp[0] = y[0]
        return x + tmp if add else x - tmp


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
        for p, s in enumerate(self.states):
            w = layout.NGWORDS + p * layout.NWORDS + layout.W_PIECE
            self.blob[w] = (self.blob[w] & ~np.uint32(1 << 17)) | (np.uint32(1 << 17) if s.dead[0] else np.uint32(0))


class state:
    """Reference: data_types/state.py:1-40 — a snapshot of one game plus a lazy per-player processor."""

    def __init__(self, backend_state, state_processor, unlocked=True):
        self.unlocked = unlocked
        self.state_processor = state_processor
        self.backend_state = backend_state.copy()
        self.is_dead = [x.dead[0] for x in backend_state.states]
        if not unlocked:
            self.lock()

    def lock(self):
        for s in self.backend_state.states:
            s.dead[0] = 1            # prevents the state from being changed by performing actions
        self.unlocked = False

    def unlock(self):
        for i, s in enumerate(self.backend_state.states):
            s.dead[0] = self.is_dead[i]
        self.unlocked = True

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            return [self.state_processor(self.backend_state, x) for x in range(*idx.indices(len(self.backend_state.states)))]
        if not hasattr(idx, "__iter__"):
            return self.state_processor(self.backend_state, idx)
        return [self.state_processor(self.backend_state, i) for i in idx]

    def __len__(self):
        return len(self.backend_state.states)

    def __iter__(self):
        self.current = -1
        return self

    def __next__(self):
        self.current += 1
        if self.current == len(self.backend_state.states):
            raise StopIteration
        return self.backend_state.states[self.current]
