"""`tetris_environment_vector` — the drop-in for the reference's worker rollout loop
(drl_tetris/worker.py:91-118), same method names and argument conventions as
environment/tetris_environment_vector.py:9-191, but ONE batched GPU environment underneath instead
of a Python list of single-game C++ handles.

    env = tetris_environment_vector(n_envs, settings={...})
    state        = env.get_state()                         # list of `state`
    reward, done = env.perform_action(actions, player=p)   # lists
    env.reset(env=[i for i, d in enumerate(done) if d])

Index conventions follow tools/utils.py:10-31 `parse_arg`: `None` = all, list/ndarray = subset,
int = broadcast.  Where the reference's vector class is broken (calls with missing arguments or
undefined names: tetris_environment_vector.py:65,91,95,112,160) the evident intent is
implemented.  Seeds: the reference seeds from time(NULL) (PythonHandle.cpp:68-71); here
`settings["seed_source"]` (a callable -> int, default wall clock seconds) is asked once per reset call.
"""
import itertools
import time

import numpy as np

from . import data_types, state_processors
from .capi import TetrisBatch

try:                                   # C helper that packs a Python list of actions (csrc/fastpack.c; built by __graft_entry__.build())
    from . import _fastpack
except ImportError:                    # not built: the Python packer below does the same, ~5x slower
    _fastpack = None
from .data_types import action, action_batch, action_list, lazy_list, maingoal_reward, null_action, snapshot_batch, state

DEFAULT_SETTINGS = {          # the env-relevant keys of experiments/presets.py:123-182
    "game_size": [22, 10],
    "pieces": [0, 1, 2, 3, 4, 5, 6],
    "n_players": 2,
    "time_elapsed_each_action": 400,
    "action_type": "place_block",
    "bar_null_moves": True,
    "state_processor": "state_dict",
    "old_state_dict": False,
    "state_processor_separate_piece": True,
    "augment_data": False,
    "extra_rewards": False,
    "reward_ammount": (1.0, 0.0),
    "render": False,
    "render_simulation": False,
    "seed_source": None,
    "field_colours": False,     # True: State.field holds tile values 1..8 like the reference (costs 120 B more state per board)
    "device": 0,
}


def parse_arg(entry_idx, data, fill_up=None, indices=False):
    """tools/utils.py:10-31"""
    if entry_idx is None:
        ret = data if type(data) is list else list(data)
        idx = list(range(len(data)))
    elif type(entry_idx) in (list, np.ndarray):
        ret = [data[i] for i in entry_idx]
        idx = entry_idx
    else:
        idx = [entry_idx] if fill_up is None else [entry_idx for _ in range(fill_up)]
        ret = [data[i] for i in idx]
    return (idx, ret) if indices else ret


class tetris_environment_vector:
    def __init__(self, n_envs, env_type=None, init_envs=None, settings=None, _lib_path=None):
        s = dict(DEFAULT_SETTINGS)
        s.update(settings or {})
        s["game_area"] = s["game_size"][0] * s["game_size"][1]      # tools/utils.py:44
        self.settings = s
        assert s["action_type"] in ("place_block", "press_key")
        self.n_envs = n_envs
        self.env_type = env_type
        self.n_players = s["n_players"]
        self.player_idxs = list(range(self.n_players))
        self.height, self.width = s["game_size"]
        self._lib_path = _lib_path
        if type(s["state_processor"]) is str:
            func, names = state_processors.func_dict[s["state_processor"]]
            self.state_processor = state_processors.state_processor(func, [s[k] for k in names])
        else:
            self.state_processor = state_processors.state_processor(s["state_processor"])
        self._seed_source = s["seed_source"] or (lambda: int(time.time()))
        seed = self._seed_source()
        self.backend = TetrisBatch(n_envs, self.n_players, self.height, self.width, pieces=s["pieces"], seeds=seed,
                                   device=s["device"], lib_path=_lib_path, colours=s["field_colours"])
        self._env_ids = list(range(n_envs))
        self._all_idx = np.arange(n_envs, dtype=np.int32)
        self._shared_zero = self._zero_reward()
        self.done = np.zeros(n_envs, bool)
        self.rounds_played = np.zeros(n_envs, np.int64)
        # running reward sums per (env, player) as numbers; the reward OBJECTS of get_info are made from them when asked for.  A
        # maingoal_reward sum only ever accumulates component 0 of what is added (reward.py:64-71), so one number per entry is
        # the whole state, plus whether a two-component reward (extra_rewards) has been added (the sum then has two components)
        self._round_sum = np.zeros((n_envs, self.n_players), np.float64)
        self._tot_sum = np.zeros((n_envs, self.n_players), np.float64)
        self._round_two = np.zeros((n_envs, self.n_players), bool)
        self._tot_two = np.zeros((n_envs, self.n_players), bool)
        self._shared_win, self._shared_loss = maingoal_reward([1]), maingoal_reward([-1])
        self._last_reward = np.full((n_envs, self.n_players), None, dtype=object)      # (an array: one assignment per step, not a loop)
        if init_envs is None or (type(init_envs) is list and all(e is None for e in init_envs)):
            self.reset()                       # "upon agreement with backend, we always reset once" (tetris_environment.py:40-41)
        else:
            self.set(init_envs)

    # ------------------------------------------------------------------ helpers
    def _zero_reward(self):
        return maingoal_reward([0])

    def _idx(self, env):
        if env is None:
            return self._all_idx
        return np.asarray(parse_arg(env, self._env_ids), dtype=np.int32)

    def _players(self, player, n):
        return [int(p) for p in parse_arg(player, self.player_idxs, fill_up=n)]

    def _who(self, player, n):
        """the acting player of each of n envs as an array (parse_arg conventions: None = every player index in turn — which,
        like the reference, only makes sense when n == n_players —, int = the same for all, list = one per env)"""
        if isinstance(player, (int, np.integer)):
            assert 0 <= player < self.n_players
            return np.full(n, int(player), np.int64)
        return np.asarray(self._players(player, n), np.int64)

    def _states_of(self, blob):
        batch = snapshot_batch(blob, self.height, self.width, self.n_players)
        processor = self.state_processor
        states = lazy_list(len(blob), lambda j: state._of_batch(batch, j, processor))
        states._batch = batch
        return states

    def _sum_objects(self, sums, two, i):
        return [maingoal_reward([sums[i, p], 0.0] if two[i, p] else [int(sums[i, p])]) for p in self.player_idxs]

    @property
    def round_reward(self):
        """[n_envs][n_players] reward objects: the sum of each player's rewards since the env's last reset"""
        return [self._sum_objects(self._round_sum, self._round_two, i) for i in range(self.n_envs)]

    @property
    def tot_reward(self):
        return [self._sum_objects(self._tot_sum, self._tot_two, i) for i in range(self.n_envs)]

    @property
    def last_reward(self):
        """reward object of each env's last perform_action, per player: [n_envs][n_players] (tetris_environment.py:111)"""
        return self._last_reward.tolist()

    def _reward(self, done, dead, player):
        """tetris_environment.reward_fcn (tetris_environment.py:135-149)"""
        base = 0
        if done:
            medead = int(dead[player])
            youdead = int(dead[1 - player]) if self.n_players > 1 else 0
            base = youdead - medead
            if medead and youdead:
                base = -1
        return base

    def _pack(self, actions, who, n):
        """n actions (Python `action` lists, or an action_batch that already holds them as arrays) of the players `who` ->
        keys uint8 [n, P, K], lens uint8 [n, P]; every other player gets the null action [0] (tetris_environment.py:106-108)"""
        who = np.asarray(who, np.int64)
        lens = np.ones((n, self.n_players), np.uint8)
        rows = np.arange(n)
        if isinstance(actions, action_batch) and "_make" in actions.__dict__:      # (a batch that was looked at is a plain list now)
            assert len(actions) == n
            keys = np.zeros((n, self.n_players, max(1, actions.keys.shape[1])), np.uint8)
            keys[rows, who] = actions.keys
            lens[rows, who] = actions.lens
            return keys, lens
        if _fastpack is not None and type(actions) is list:
            width = _fastpack.max_len(actions, action)              # (TypeError for an element that is not an `action`)
            assert width <= 255, "an action may hold at most 255 keys (uint8 length on the device)"
            keys = np.zeros((n, self.n_players, max(1, width)), np.uint8)
            _fastpack.fill(actions, who.ctypes.data, keys.ctypes.data, lens.ctypes.data, n, self.n_players, keys.shape[2])
            return keys, lens
        assert set(map(type, actions)) <= {action}, "perform_action(action a, int p) was called with an action that is not of type `action`"
        packed = list(map(bytes, actions))          # (a key outside 0..255 raises here)
        length = np.fromiter(map(len, packed), np.int64, n)
        width = int(length.max(initial=0))
        assert width <= 255, "an action may hold at most 255 keys (uint8 length on the device)"
        keys = np.zeros((n, self.n_players, max(1, width)), np.uint8)
        flat = np.frombuffer(b"".join(packed), np.uint8)
        r = np.repeat(rows, length)
        keys[r, who[r], np.arange(len(flat)) - np.repeat(np.cumsum(length) - length, length)] = flat
        lens[rows, who] = length
        return keys, lens

    # ------------------------------------------------------------------ env interface
    def reset(self, env=None):
        idx = self._idx(env)
        if len(idx) == 0:
            return []
        self.backend.reset(idx, seeds=self._seed_source())
        self.done[idx] = False
        self.rounds_played[idx] += 1
        self._round_sum[idx] = 0.0
        self._round_two[idx] = False
        return [None] * len(idx)

    def perform_action(self, actions, env=None, player=None):
        idx = self._idx(env)
        n = len(idx)
        who = self._who(player, n)
        if not isinstance(actions, action_batch):
            actions = list(actions)
        assert len(actions) == n and len(who) == n
        keys, lens = self._pack(actions, who, n)
        done, _lines, dead = self.backend.step_keys(keys, lens, ms=self.settings["time_elapsed_each_action"], idx=None if env is None else idx)
        # reward_fcn (tetris_environment.py:135-149) is 0 unless the round just ended: only those envs get a reward object of
        # their own and an update of the running sums (x + 0 leaves the sums as they are); the others share one zero reward
        # (reward objects are never modified in place: `+` / `-` return new ones).  Nothing below loops over all n envs.
        done_b = done.astype(bool)
        self.done[idx] = done_b
        zero = self._shared_zero
        rewards = [zero] * n
        self._last_reward[idx, who] = zero
        if self.settings["extra_rewards"]:
            # tetris_environment.py:144-149: two components, [w_base * win/lose signal, w_combo * my combo count], every step
            w_base, w_combo = self.settings["reward_ammount"]
            combo = self.backend.observe_packed(None if env is None else idx, who.astype(np.uint8))[1][0][:, 4]
            for j in range(n):
                i, p = idx[j], int(who[j])
                r = maingoal_reward([w_base * self._reward(bool(done_b[j]), dead[j], p), w_combo * int(combo[j])])
                rewards[j] = self._last_reward[i, p] = r
                self._round_sum[i, p] += r.extrinsic[0]
                self._tot_sum[i, p] += r.extrinsic[0]
            self._round_two[idx, who] = True
            self._tot_two[idx, who] = True
        else:
            dj = np.nonzero(done_b)[0]
            if len(dj):
                # reward_fcn for the envs whose round ended, all at once: you dead - me dead, -1 when both are
                me = dead[dj, who[dj]].astype(np.int64)
                you = dead[dj, 1 - who[dj]].astype(np.int64) if self.n_players > 1 else np.zeros(len(dj), np.int64)
                base = np.where((me == 1) & (you == 1), -1, you - me)
                np.add.at(self._round_sum, (idx[dj], who[dj]), base)
                np.add.at(self._tot_sum, (idx[dj], who[dj]), base)
                for j, v in zip(dj[base != 0].tolist(), base[base != 0].tolist()):      # (the +-1 reward objects are shared too)
                    rewards[j] = self._last_reward[idx[j], who[j]] = self._shared_win if v > 0 else self._shared_loss
        return rewards, done_b.tolist()

    def get_state(self, env=None):
        """One `state` per env (data_types/state.py:1-40), cut from ONE snapshot of the listed games: a single kernel and a single
        copy for all of them; the Python objects and the State views are made when they are first looked at (data_types.lazy_list,
        snapshot_batch)."""
        return self._states_of(self.backend.snapshot(None if env is None else self._idx(env)))

    def set(self, target, env=None):
        """Restore games from state / backend_snapshot / another vector env (tetris_environment.py:168-176)."""
        idx = self._idx(env)
        if isinstance(target, tetris_environment_vector):
            src = target.backend.snapshot(np.arange(len(idx), dtype=np.int32) if env is None else idx)
            self.backend.restore(src, idx)
            return [None for _ in idx]
        if isinstance(target, lazy_list) and "_make" in target.__dict__ and getattr(target, "_batch", None) is not None and len(target) == len(idx):
            batch = target._batch                        # states nobody has looked at: their words go back as they came
            self.backend.restore(batch.blob, idx)
            self.done[idx] = ((batch.blob[:, data_types.layout.G_META] >> 16) & 1).astype(bool)
            return [None for _ in idx]
        targets = target if isinstance(target, list) else [target for _ in idx]
        blobs = np.zeros((len(idx), self.backend.snapshot_words), np.uint32)
        for j, t in enumerate(targets):
            b = t.backend_state if isinstance(t, state) else t
            if not isinstance(b, data_types.backend_snapshot):
                raise TypeError(f"cannot set an environment from {type(t)}")
            b.sync_dead_to_blob()
            blobs[j] = b.blob
        self.backend.restore(blobs, idx)
        self.done[idx] = [bool(t.backend_state.round_over if isinstance(t, state) else t.round_over) for t in targets]
        return [None for _ in idx]

    def simulate_actions(self, actions, env=None, player=None, finalize=True):
        """Per env: the states reached by each action of its action_list from the CURRENT state, which is
        left untouched (tetris_environment.simulate_actions, tetris_environment.py:87-100).  All
        (env, action) pairs are stepped in one scratch batch."""
        idx = self._idx(env)
        players = self._players(player, len(idx))
        counts = [len(a) for a in actions]
        total = sum(counts)
        if total == 0:
            return [[] for _ in idx]
        anchors = self.backend.snapshot(idx)
        scratch = TetrisBatch(total, self.n_players, self.height, self.width, pieces=self.settings["pieces"], seeds=0,
                              device=self.settings["device"], lib_path=self._lib_path, colours=self.settings["field_colours"])
        scratch.restore(np.repeat(anchors, counts, axis=0))
        flat_actions = [a if type(a) is action else action(a) for al in actions for a in al]
        flat_players = np.repeat(np.asarray(players, np.int64), counts)
        keys, lens = self._pack(flat_actions, flat_players, total)
        if finalize:
            scratch.step_keys(keys, lens, ms=self.settings["time_elapsed_each_action"])
        else:
            scratch.make_actions(keys, lens)
        batch = snapshot_batch(scratch.snapshot(), self.height, self.width, self.n_players)
        scratch.close()
        out, k = [], 0
        for c in counts:
            out.append([state._of_batch(batch, k + j, self.state_processor) for j in range(c)])
            k += c
        return out

    def get_actions(self, env=None, player=None):
        idx = self._idx(env)
        players = self._players(player, len(idx))
        assert len(idx) == len(players), "You want to specify one player per tetris_environment that the tetris_environment_vector manages"
        if self.settings["action_type"] == "press_key":
            return [action_list([[k] for k in range(11)]) for _ in idx]      # tetris_environment.py:85-86
        lists = self.backend.get_actions(idx, players)
        return [action_list(l, remove_null=self.settings["bar_null_moves"]) for l in lists]

    def get_random_action(self, env=None, player=None):
        lists = self.get_actions(env=env, player=player)
        return [al[np.random.randint(low=0, high=len(al))] for al in lists]

    def simulate_all_actions(self, env=None, player=None, finalize=True):
        return self.simulate_actions(self.get_actions(env=env, player=player), env=env, player=player, finalize=finalize)

    def get_winner(self, env=None, player=None):
        idx = self._idx(env)
        rec, ro, _ = self.backend.observe(idx)
        out = []
        for j in range(len(idx)):
            if not ro[j]:
                out.append(None)
                continue
            alive = [p for p in self.player_idxs if not rec[j, p]["dead"]]
            out.append(alive[0] if alive else 666)        # tetris_environment.py:118-125
        return out

    def get_info(self, env=None):
        idx = self._idx(env)
        rec, _, _ = self.backend.observe(idx)
        last = self.last_reward
        return [{"is_dead": [rec[j, p]["dead"] for p in self.player_idxs], "reward": last[i],
                 "tot_reward": self._sum_objects(self._tot_sum, self._tot_two, i),
                 "round_reward": self._sum_objects(self._round_sum, self._round_two, i),
                 "rounds_played": int(self.rounds_played[i])} for j, i in enumerate(idx)]

    def get_fields(self, env=None):
        rec, _, _ = self.backend.observe(self._idx(env))
        return [[rec[j, p]["field"][: self.height, : self.width] for p in self.player_idxs] for j in range(len(rec))]

    def render(self, env=None):
        return None       # visualisation (env_utils/draw_tetris.py) is out of scope; kept as a no-op hook

    def copy(self):
        return tetris_environment_vector(self.n_envs, self.env_type, init_envs=self, settings=self.settings, _lib_path=self._lib_path)

    def generate_pieces(self, env=None):
        p = self.settings["pieces"]
        return [(p * 7)[:7] for _ in self._idx(env)]

    # ------------------------------------------------------------------ pickling (tetris_environment_vector.py:179-191)
    def __getstate__(self):
        """Everything but the device handle: settings, reward bookkeeping and the games as snapshot words.  Unlike the
        reference's pickle (PythonHandle.h:180-182 drops the generators) the RNG positions survive.  A `seed_source` that
        cannot be pickled (a lambda) is replaced by the default wall-clock source on load."""
        d = {k: v for k, v in self.__dict__.items() if k not in ("backend", "state_processor", "_seed_source")}
        d["settings"] = dict(self.settings)
        try:
            import pickle
            pickle.dumps(self.settings["seed_source"])
        except Exception:
            d["settings"]["seed_source"] = None
        if type(d["settings"]["state_processor"]) is not str:
            raise TypeError("a vector env with a custom state_processor callable cannot be pickled; name it in state_processors.func_dict")
        d["_games"] = self.backend.snapshot()
        return d

    def __setstate__(self, d):
        games = d.pop("_games")
        # the constructor draws seeds (batch creation + the initial reset): build with a throw-away source so that the
        # unpickled seed source continues exactly where the pickled one stood
        fresh = tetris_environment_vector(d["n_envs"], d["env_type"], settings=dict(d["settings"], seed_source=lambda: 0),
                                          _lib_path=d["_lib_path"])
        self.__dict__.update(fresh.__dict__)
        self.__dict__.update(d)
        self._seed_source = self.settings["seed_source"] or (lambda: int(time.time()))
        self.backend.restore(games)

    def __str__(self, env=None):
        width = max(len(k) for k in self.settings)
        body = "".join("\t{:{}}\t{}\n".format(k, width, v) for k, v in self.settings.items())
        return "<tetris_vector_env>" + "".join("tetris_environment settings:\n" + body for _ in self._idx(env)) + "</tetris_vector_env>"


class tetris_environment:
    """Single-game environment with the reference's method signatures (environment/tetris_environment.py:11-227), as the
    agents use it for their `sandbox` (agents/sherlock_agent/sherlock_agent.py:94; agents/sherlock_agent/sherlock_utils.py:13-20;
    agents/random_agent.py:31-35): `get_actions(state, player)`, `simulate_actions`, `simulate_all_actions(state, player)`,
    `perform_action(action, player) -> (reward, done)`, `get_state()`, `set`, `copy`, `reset`.  A 1-game batch underneath."""

    def __init__(self, id=None, settings=None, init_env=None, _lib_path=None):
        self.id = id
        src = init_env._vec if isinstance(init_env, tetris_environment) else init_env
        self._vec = tetris_environment_vector(1, type(self), init_envs=src, settings=settings, _lib_path=_lib_path)
        self.settings = self._vec.settings
        self.player_idxs = self._vec.player_idxs
        self.state_processor = self._vec.state_processor

    @property
    def done(self):
        return bool(self._vec.done[0])

    @property
    def backend(self):
        return self._vec.backend

    def reset(self):
        self._vec.reset()

    def get_state(self):
        return self._vec.get_state()[0]

    def set(self, e):
        if isinstance(e, tetris_environment):
            e = e._vec
        self._vec.set(e if isinstance(e, tetris_environment_vector) else [e])

    def copy(self):
        return tetris_environment(settings=self.settings, init_env=self, _lib_path=self._vec._lib_path)

    def get_actions(self, state, player=None):
        assert type(player) is int, f"tetris_environment.get_actions(int player) was called with type(player)={type(player)}"
        self.set(state)                               # tetris_environment.py:80
        return self._vec.get_actions(player=player)[0]

    def get_random_action(self, player=None):
        return self._vec.get_random_action(player=player)[0]

    def simulate_actions(self, actions, player=None, finalize=True):
        assert type(actions) is action_list, f"simulate_actions was called with type(actions)={type(actions)}"
        return self._vec.simulate_actions([actions], player=player, finalize=finalize)[0]

    def simulate_all_actions(self, state, player=None, finalize=True):
        return self.simulate_actions(self.get_actions(state, player=player), player=player, finalize=finalize)

    def perform_action(self, action, player=None, simulate=False, finalize=True):
        assert type(player) is int, f"tetris_environment.perform_action(action a,int p) was called with type(player)={type(player)}"
        if finalize and not simulate:
            r, d = self._vec.perform_action([action], player=player)
            return r[0], d[0]
        keys, lens = self._vec._pack([action], [player], 1)      # simulate=True: no reward bookkeeping (tetris_environment.py:109-116)
        if finalize:
            done, _, _ = self._vec.backend.step_keys(keys, lens, ms=self.settings["time_elapsed_each_action"])
            self._vec.done[0] = bool(done[0])
        else:
            self._vec.backend.make_actions(keys, lens)
        return None, self.done

    def get_winner(self):
        return self._vec.get_winner()[0]

    def get_info(self):
        return self._vec.get_info()[0]

    def get_fields(self):
        return self._vec.get_fields()[0]

    def render(self):
        return None

    def generate_pieces(self):
        return self._vec.generate_pieces()[0]

    def __str__(self):
        width = max(len(k) for k in self.settings)
        return "tetris_environment settings:\n" + "".join("\t{:{}}\t{}\n".format(k, width, v) for k, v in self.settings.items())
