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
    """What the reference's pybind `State` exposes for one player (PythonHandle.h:54-82), as numpy arrays."""

    __slots__ = ("field", "piece", "x", "y", "inc_lines", "combo_time", "combo_count", "nextpiece", "reward", "dead")


# piece grids as State.piece shows them: SURVEY App. B row masks (rows y0..y3 as hex nibbles, bit x = column x) and the grid values
# {4,3,5,7,2,1,6} of gamePlay.cpp:125-139; index [kind][current rotation]; kind 7 = no piece
_SHAPE_ROWS = ((0x2260, 0x0710, 0x3220, 0x4700), (0x2230, 0x1700, 0x6220, 0x0740), (0x2640, 0x0630, 0x1320, 0x6300),
               (0x2310, 0x3600, 0x4620, 0x0360), (0x2222, 0x0f00, 0x4444, 0x00f0), (0x0720, 0x2320, 0x2700, 0x2620),
               (0x0660, 0x0660, 0x0660, 0x0660))
_GRID_VALUE = (4, 3, 5, 7, 2, 1, 6)
_GRIDS = np.zeros((8, 4, 4, 4), np.uint8)
for _k in range(7):
    for _r in range(4):
        for _gy in range(4):
            for _gx in range(4):
                if (_SHAPE_ROWS[_k][_r] >> (4 * (3 - _gy) + _gx)) & 1:
                    _GRIDS[_k, _r, _gy, _gx] = _GRID_VALUE[_k]


class snapshot_batch:
    """The games of ONE get_state() / simulate_actions() call as they came off the device in one call: snapshot words [n, words]
    (csrc/tetris_layout.h; the exact state incl. RNG position, what tetris_restore takes back).  Everything the reference's
    `State` views show (PythonHandle.h:54-82) is a pure function of these words and is decoded here, for all n games at once,
    the first time any of it is looked at; a loop that only carries states along never pays for it."""

    def __init__(self, blob, height, width, n_players):
        self.blob = blob
        self.height, self.width, self.n_players = int(height), int(width), int(n_players)
        self.nw = (blob.shape[1] - layout.NGWORDS) // self.n_players
        self._views = None

    def __len__(self):
        return len(self.blob)

    def state_dicts(self, parameters):
        """the built-in `state_dict` processor for every game and player of the batch (state_processors.state_dict_batch), cached"""
        if getattr(self, "_dicts_key", None) is not parameters:       # (a processor's parameter list is one object for its lifetime)
            from . import state_processors
            self._dicts, self._dicts_key = state_processors.state_dict_batch(self, parameters), parameters
        return self._dicts

    def views(self):
        if self._views is None:
            P, H, nw = self.n_players, self.height, self.nw
            boards = self.blob[:, layout.NGWORDS:].reshape(len(self.blob), P, nw)
            def bit(cols):                           # column words [n,P,10] -> cells [n,P,H,10]: bit y of column c = cell (y, c)
                b = np.unpackbits(np.ascontiguousarray(cols).view(np.uint8).reshape(len(cols), P, layout.NCOL, 4), axis=3, bitorder="little")
                return np.ascontiguousarray(b[:, :, :, :H].transpose(0, 1, 3, 2))
            field = bit(boards[:, :, layout.W_COL0:layout.W_COL0 + layout.NCOL])
            if nw > layout.NWORDS:                       # colour planes: cell value = 1 + 3-bit plane value on occupied squares
                t0 = layout.NWORDS
                value = 1 + bit(boards[:, :, t0:t0 + 10]) + 2 * bit(boards[:, :, t0 + 10:t0 + 20]) + 4 * bit(boards[:, :, t0 + 20:t0 + 30])
                field = field * value
            w, m, dc = boards[:, :, layout.W_PIECE], boards[:, :, layout.W_MISC], boards[:, :, layout.W_DROPCOMBO]
            meta = self.blob[:, layout.G_META]
            self._views = {
                "field": field, "piece": _GRIDS[w & 7, (w >> 3) & 3],
                "x": (((w >> 5) & 15).astype(np.int16) - 4).astype(np.int8), "y": ((w >> 9) & 31).astype(np.int8),
                "nextpiece": ((w >> 14) & 7).astype(np.uint8), "dead": ((w >> 17) & 1).astype(np.uint8),
                "reward": ((w >> 19) & 255).astype(np.uint8), "inc_lines": (m & 255).astype(np.uint8),
                "combo_count": ((m >> 8) & 255).astype(np.uint8), "combo_time": (dc >> 16).astype(np.uint16),
                "round_over": ((meta >> 16) & 1).astype(np.uint8), "last_winner": (((meta >> 17) & 15).astype(np.int16) - 1).astype(np.int8),
            }
        return self._views


class backend_snapshot:
    """Stand-in for the copied PythonHandle inside a reference `state`: `.states[p]` views plus the raw snapshot words that
    restore the exact game (RNG position included) on the GPU.  A view into game `j` of a snapshot_batch; the words are copied
    out of the batch, and the views decoded, when first asked for."""

    def __init__(self, batch, j):
        self._batch, self._j = batch, int(j)
        self._blob = self._states = None
        self.height, self.width = batch.height, batch.width

    @classmethod
    def from_blob(cls, blob, height, width, n_players):
        """a standalone snapshot of one game from its words (unpickling, tests)"""
        return cls(snapshot_batch(np.array(blob, dtype=np.uint32, copy=True).reshape(1, -1), height, width, n_players), 0)

    @property
    def blob(self):
        if self._blob is None:
            self._blob = self._batch.blob[self._j].copy()
        return self._blob

    @property
    def states(self):
        if self._states is None:
            v, j = self._batch.views(), self._j
            out = []
            for p in range(self._batch.n_players):
                s = _player_view()
                s.field, s.piece = v["field"][j, p], v["piece"][j, p]
                for name in ("x", "y", "inc_lines", "combo_time", "combo_count", "nextpiece", "reward"):
                    setattr(s, name, v[name][j, p:p + 1])
                s.dead = v["dead"][j, p:p + 1].copy()      # the one view Python writes (state.py:11,16): never shared
                out.append(s)
            self._states = out
        return self._states

    @property
    def round_over(self):
        return int((self._batch.blob[self._j, layout.G_META] >> 16) & 1)

    @property
    def last_winner(self):
        return int((self._batch.blob[self._j, layout.G_META] >> 17) & 15) - 1

    @property
    def n_players(self):
        return self._batch.n_players

    @property
    def records(self):
        """per-player scalars beyond the State views (time_ms, draw counters), decoded from the words: a structured array [P]"""
        b = self.blob[layout.NGWORDS:].reshape(self._batch.n_players, self._batch.nw)
        rec = np.zeros(self._batch.n_players, dtype=[("time_ms", np.int32), ("piece_draws", np.uint32), ("hole_draws", np.uint32),
                                                     ("dead", np.uint8), ("next", np.uint8), ("reward", np.uint8)])
        rec["time_ms"] = b[:, layout.W_TIME].view(np.int32)
        rec["piece_draws"], rec["hole_draws"] = b[:, layout.W_PIECE_DRAWS], b[:, layout.W_HOLE_DRAWS]
        rec["dead"], rec["next"], rec["reward"] = (b[:, layout.W_PIECE] >> 17) & 1, (b[:, layout.W_PIECE] >> 14) & 7, (b[:, layout.W_PIECE] >> 19) & 255
        return rec

    def dead_flags(self):
        if self._states is not None:
            return [s.dead[0] for s in self._states]
        nw, w = self._batch.nw, self._batch.blob[self._j]
        return [np.uint8((w[layout.NGWORDS + p * nw + layout.W_PIECE] >> 17) & 1) for p in range(self._batch.n_players)]

    def copy(self):
        b = backend_snapshot(self._batch, self._j)
        if self._blob is not None:
            b._blob = self._blob.copy()
        if self._states is not None:
            for mine, theirs in zip(b.states, self._states):
                mine.dead[0] = theirs.dead[0]
        return b

    def sync_dead_to_blob(self):
        """Python may write `states[p].dead[0]` (state.py:11,16); carry it into the restore words."""
        if self._states is None:
            return
        nw = self._batch.nw          # words per player-board: 39, or 69 for batches with colour planes (tetris_layout.h)
        for p, s in enumerate(self._states):
            w = layout.NGWORDS + p * nw + layout.W_PIECE
            self.blob[w] = (self.blob[w] & ~np.uint32(1 << 17)) | (np.uint32(1 << 17) if s.dead[0] else np.uint32(0))

    def __reduce__(self):            # pickles as its own words (not as the batch it was cut from)
        self.sync_dead_to_blob()
        return (backend_snapshot.from_blob, (self.blob, self.height, self.width, self._batch.n_players))


class state:
    """A snapshot of one game plus a lazy per-player processor — same surface as the reference's data_types/state.py:1-40
    (`backend_state`, `is_dead`, `lock`/`unlock`, indexing by player, iteration over the raw per-player views)."""

    def __init__(self, backend_state, state_processor, unlocked=True):
        self.state_processor = state_processor
        self.backend_state = backend_state.copy()
        self.is_dead = self.backend_state.dead_flags()
        self.unlocked = True
        if not unlocked:
            self.lock()

    @classmethod
    def _of_batch(cls, batch, j, state_processor):
        """game j of a snapshot_batch, without touching it (what get_state hands out)"""
        s = cls.__new__(cls)
        s.state_processor, s.unlocked = state_processor, True
        s._pending = (batch, j)
        return s

    def __getattr__(self, name):     # only reached for attributes not set yet: the two lazy ones of _of_batch
        if name in ("backend_state", "is_dead"):
            pending = self.__dict__.pop("_pending", None)
            if pending is not None:
                self.backend_state = backend_snapshot(*pending)
                self.is_dead = self.backend_state.dead_flags()
                return self.__dict__[name]
        raise AttributeError(name)

    def __getstate__(self):
        self.backend_state           # (materialise before pickling)
        return self.__dict__

    def __setstate__(self, d):
        self.__dict__.update(d)

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

    def _process(self, p):
        # the built-in state_dict of a state that still is a view into its snapshot batch: from the batch's arrays (all games at
        # once, the first time any is asked for); anything else — a custom processor, a state whose views were written to —
        # through the processor itself
        proc = self.state_processor
        if getattr(proc.func, "__name__", "") == "state_dict":
            pending = self.__dict__.get("_pending")
            if pending is None:
                b = self.__dict__.get("backend_state")
                pending = (b._batch, b._j) if b is not None and b._states is None else None
            if pending is not None and isinstance(p, (int, np.integer)) and 0 <= p < pending[0].n_players:
                return pending[0].state_dicts(proc.parameters).item(pending[1], int(p))
        return proc(self.backend_state, p)

    def __getitem__(self, idx):
        players = self._players(idx)
        if players is None:
            return self._process(idx)
        return [self._process(p) for p in players]

    def __len__(self):
        return self.backend_state.n_players

    def __iter__(self):
        return iter(self.backend_state.states)


class lazy_list(list):
    """A list whose n elements are made by `make(j)` the first time ANY of them is looked at (or the list is changed): what
    get_state returns for thousands of games — a worker loop that hands the states on (to an experience buffer, to the next
    get_state) does not pay for n Python objects per call, code that indexes, iterates or pickles it sees an ordinary list."""

    def __init__(self, n, make):
        super().__init__()
        self._n, self._make = int(n), make

    def _fill(self):
        make = self.__dict__.pop("_make", None)
        if make is not None:
            list.extend(self, [make(j) for j in range(self._n)])

    def __len__(self):
        return self._n if "_make" in self.__dict__ else list.__len__(self)

    def __reduce_ex__(self, protocol):
        self._fill()
        return (list, (list(self),))

    def __radd__(self, other):       # plain_list + lazy_list: list.__add__ would read this object's (still empty) storage directly
        self._fill()
        return list.__add__(other, self)


def _filled(name):
    def method(self, *args, **kwargs):
        self._fill()
        return getattr(list, name)(self, *args, **kwargs)
    method.__name__ = name
    return method


for _name in ("__getitem__", "__iter__", "__contains__", "__reversed__", "__eq__", "__ne__", "__lt__", "__le__", "__gt__", "__ge__",
              "__repr__", "__add__", "__mul__", "__rmul__", "__iadd__", "__imul__", "__setitem__", "__delitem__", "index", "count",
              "copy", "append", "extend", "insert", "pop", "remove", "sort", "reverse", "clear"):
    setattr(lazy_list, _name, _filled(_name))
lazy_list.__hash__ = None


class action_batch(lazy_list):
    """n actions held as arrays (keys uint8 [n, K], lens uint8 [n]): what an agent that decides for all envs at once hands to
    perform_action — the environment takes the arrays as they are instead of packing n Python lists; looked at as a list it
    holds `action` objects like any other."""

    def __init__(self, keys, lens):
        self.keys = np.ascontiguousarray(keys, dtype=np.uint8)
        self.lens = np.ascontiguousarray(lens, dtype=np.uint8)
        assert self.keys.ndim == 2 and self.lens.shape == (len(self.keys),)
        super().__init__(len(self.keys), lambda j: action(self.keys[j, : self.lens[j]].tolist()))

    _rt_tables = {}

    @classmethod
    def from_rt(cls, rot, trans):
        """SVENton's (rotation, translation) encoding [8]*r + [2] + [3]*t + [7] (sventon_utils.py:9-13) for n envs"""
        rot, trans = np.asarray(rot, np.int64).reshape(-1), np.asarray(trans, np.int64).reshape(-1)
        if len(rot) == 0:
            return cls(np.zeros((0, 2), np.uint8), np.zeros(0, np.uint8))
        if rot.min() < 0 or trans.min() < 0:
            raise ValueError("rotation and translation counts must be >= 0")
        # the key list depends on (r, t) only: one table row per pair that can occur, then a gather for the n envs
        n_r, n_t = int(rot.max()) + 1, int(trans.max()) + 1
        tables = cls._rt_tables.get((n_r, n_t))
        if tables is None:
            r, t = np.divmod(np.arange(n_r * n_t), n_t)
            col = np.arange(n_r + n_t)[None, :]
            keys = np.where(col < r[:, None], 8, np.where(col == r[:, None], 2, np.where(col <= (r + t)[:, None], 3, 7)))
            keys = np.where(col > (r + t + 1)[:, None], 0, keys).astype(np.uint8)
            tables = cls._rt_tables[(n_r, n_t)] = (keys, (r + t + 2).astype(np.uint8))
        code = rot * n_t + trans
        return cls(np.take(tables[0], code, axis=0), np.take(tables[1], code))
