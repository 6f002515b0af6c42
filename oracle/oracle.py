"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes front-end of ``oracle/liboracle.so`` (the CPU restatement in
``tetris_oracle.c``) plus a loader for the compiled reference in ``oracle/_ref``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker.  The product package
(``drl-tetris_amd/``) never does.
"""
import ctypes as C
import glob
import importlib.util
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MAX_H, W, FIFO_CAP = 32, 10, 16

# mirrors `struct or_record` in tetris_oracle.h field for field (C layout, aligned)
RECORD = np.dtype(
    [
        ("field", np.uint8, (MAX_H, W)),
        ("grid", np.uint8, (4, 4)),
        ("x", np.int8),
        ("y", np.int8),
        ("piece", np.uint8),
        ("tile", np.uint8),
        ("spawn_rot", np.uint8),
        ("cur_rot", np.uint8),
        ("big", np.uint8),
        ("next", np.uint8),
        ("dead", np.uint8),
        ("reward", np.uint8),
        ("inc_count", np.uint8),
        ("combo_count", np.uint8),
        ("combo_remaining", np.uint16),
        ("lock_armed", np.uint8),
        ("fifo_len", np.uint8),
        ("line_count", np.uint8),
        ("fifo_overflow", np.uint8),
        ("time_ms", np.int32),
        ("incoming", np.float32),
        ("drop_delay", np.int32),
        ("drop_time", np.int32),
        ("speedup_time", np.int32),
        ("lock_time", np.int32),
        ("min_remaining", np.int32),
        ("combo_start", np.int32),
        ("combo_time", np.int32),
        ("fifo_delay", np.int32, (FIFO_CAP,)),
        ("fifo_count", np.int16, (FIFO_CAP,)),
        ("lines_sent", np.uint16),
        ("lines_cleared", np.uint16),
        ("lines_blocked", np.uint16),
        ("garbage_cleared", np.uint16),
        ("max_combo", np.uint16),
        ("lines_cleared_seen", np.uint16),
        ("weights", np.float32, (7,)),
        ("piece_draws", np.uint32),
        ("hole_draws", np.uint32),
    ],
    align=True,
)

_lib = None


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(os.path.join(HERE, "liboracle.so")):
        subprocess.run(["make", "-C", HERE, "liboracle.so"], check=True, capture_output=True)
    if os.path.isdir("/root/reference") and (force or not glob.glob(os.path.join(HERE, "_ref", "tetris_env*.so"))):
        subprocess.run(["make", "-C", HERE, "ref"], check=True, capture_output=True)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = os.path.join(HERE, "liboracle.so")
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(HERE, "tetris_oracle.c")):
        subprocess.run(["make", "-C", HERE, "liboracle.so"], check=True, capture_output=True)
    L = C.CDLL(path)
    vp, i32, u8p = C.c_void_p, C.c_int, C.c_void_p
    L.or_record_size.restype = i32
    L.or_create.restype = vp
    L.or_create.argtypes = [i32, i32, i32, i32, vp, vp]
    L.or_destroy.argtypes = [vp]
    L.or_reset.argtypes = [vp, vp, i32, vp]
    L.or_make_actions.argtypes = [vp, vp, i32, u8p, u8p, i32]
    L.or_finish_actions.argtypes = [vp, vp, i32, i32, u8p]
    L.or_step_rt.argtypes = [vp, u8p, u8p, u8p, i32, u8p]
    L.or_observe.argtypes = [vp, vp, i32, vp, vp, vp]
    L.or_copy_games.argtypes = [vp, vp, vp, vp, i32]
    L.or_set_dead.argtypes = [vp, vp, i32, vp]
    L.or_enumerate_drops.argtypes = [vp, vp, i32, vp, vp, vp, vp, vp]
    L.or_rollout_random.argtypes = [vp, C.c_uint32, C.c_uint64, i32, i32, vp, vp, i32]
    L.or_rollout_random_shard.argtypes = [vp, C.c_uint32, C.c_uint64, i32, i32, vp, vp, i32, C.c_uint32]
    L.or_mt19937_block.argtypes = [C.c_uint32, vp, i32]
    L.or_philox4x32_10.argtypes = [C.c_uint32] * 6 + [vp]
    L.or_combo_pow.restype = C.c_double
    L.or_combo_pow.argtypes = [i32]
    if hasattr(L, "or_get_actions"):
        L.or_get_actions.restype = i32
        L.or_get_actions.argtypes = [vp, i32, i32, vp, vp, i32, i32]
    assert L.or_record_size() == RECORD.itemsize, (L.or_record_size(), RECORD.itemsize)
    _lib = L
    return L


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _idx(idx, n_all):
    if idx is None:
        return None, n_all
    a = np.ascontiguousarray(idx, dtype=np.int32)
    return a, len(a)


class OracleBatch:
    """N independent games (each = one reference ``PythonHandle``) on the CPU."""

    def __init__(self, n_games, n_players=2, height=20, width=10, pieces=(0, 1, 2, 3, 4, 5, 6), seeds=None):
        self.L = lib()
        self.n_games, self.n_players, self.height, self.width = n_games, n_players, height, width
        pm = np.array((list(pieces) * 7)[:7], dtype=np.uint8)  # tetris_environment.py:191-193
        s = None if seeds is None else np.ascontiguousarray(np.broadcast_to(np.asarray(seeds), (n_games,)).astype(np.int64).astype(np.int16))
        self.h = self.L.or_create(n_games, n_players, height, width, _p(pm), _p(s))
        if not self.h:
            raise ValueError("or_create rejected the configuration")

    def __del__(self):
        if getattr(self, "h", None):
            self.L.or_destroy(self.h)
            self.h = None

    def reset(self, idx=None, seeds=0):
        a, n = _idx(idx, self.n_games)
        s = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds), (n,)).astype(np.int64).astype(np.int16))
        self.L.or_reset(self.h, _p(a), n, _p(s))

    def make_actions(self, keys, lens, idx=None):
        """keys uint8 [n, P, K], lens uint8 [n, P]"""
        a, n = _idx(idx, self.n_games)
        keys = np.ascontiguousarray(keys, dtype=np.uint8)
        lens = np.ascontiguousarray(lens, dtype=np.uint8)
        assert keys.shape[:2] == (n, self.n_players) and lens.shape == (n, self.n_players)
        self.L.or_make_actions(self.h, _p(a), n, _p(keys), _p(lens), keys.shape[2])

    def finish_actions(self, ms=400, idx=None):
        a, n = _idx(idx, self.n_games)
        done = np.zeros(n, dtype=np.uint8)
        self.L.or_finish_actions(self.h, _p(a), n, ms, _p(done))
        return done

    def step_rt(self, rot, trans, player=None, ms=400):
        rot = np.ascontiguousarray(rot, dtype=np.uint8)
        trans = np.ascontiguousarray(trans, dtype=np.uint8)
        pl = np.zeros(self.n_games, np.uint8) if player is None else np.ascontiguousarray(np.broadcast_to(player, (self.n_games,)), dtype=np.uint8)
        done = np.zeros(self.n_games, dtype=np.uint8)
        self.L.or_step_rt(self.h, _p(rot), _p(trans), _p(pl), ms, _p(done))
        return done

    def observe(self, idx=None):
        a, n = _idx(idx, self.n_games)
        rec = np.zeros((n, self.n_players), dtype=RECORD)
        ro = np.zeros(n, dtype=np.uint8)
        lw = np.zeros(n, dtype=np.int8)
        self.L.or_observe(self.h, _p(a), n, _p(rec), _p(ro), _p(lw))
        return rec, ro, lw

    def copy_from(self, src, dst_idx=None, src_idx=None):
        d, n = _idx(dst_idx, self.n_games)
        s, n2 = _idx(src_idx, src.n_games)
        assert n == n2
        self.L.or_copy_games(self.h, _p(d), src.h, _p(s), n)

    def set_dead(self, dead, idx=None):
        a, n = _idx(idx, self.n_games)
        dead = np.ascontiguousarray(dead, dtype=np.uint8).reshape(n, self.n_players)
        self.L.or_set_dead(self.h, _p(a), n, _p(dead))

    def enumerate_drops(self, idx=None, player=None, cells=True):
        """-> valid u8 [n,4,10], land_y i8 [n,4,10], cleared u8 [n,4,10], after u8 [n,4,10,H,W] (or None)"""
        a, n = _idx(idx, self.n_games)
        pl = None if player is None else np.ascontiguousarray(np.broadcast_to(player, (n,)), dtype=np.uint8)
        valid = np.zeros((n, 4, 10), np.uint8)
        land = np.zeros((n, 4, 10), np.int8)
        cleared = np.zeros((n, 4, 10), np.uint8)
        after = np.zeros((n, 4, 10, self.height, self.width), np.uint8) if cells else None
        self.L.or_enumerate_drops(self.h, _p(a), n, _p(pl), _p(valid), _p(land), _p(cleared), _p(after))
        return valid, land, cleared, after

    def get_actions(self, game, player, max_lists=256, max_keys=64):
        keys = np.zeros((max_lists, max_keys), np.uint8)
        lens = np.zeros(max_lists, np.uint8)
        n = self.L.or_get_actions(self.h, game, player, _p(keys), _p(lens), max_lists, max_keys)
        assert n <= max_lists
        return [keys[i, : lens[i]].tolist() for i in range(n)]

    def rollout_random(self, steps, policy_seed=0xD71, first_step=0, ms=400, episode=None, threads=1, game_offset=0):
        ep = np.zeros(self.n_games, np.uint32) if episode is None else episode
        counters = np.zeros(4, np.uint64)
        self.L.or_rollout_random_shard(self.h, policy_seed, first_step, steps, ms, _p(ep), _p(counters), threads, game_offset)
        return ep, counters


def mt19937(seed, n):
    out = np.zeros(n, np.uint32)
    lib().or_mt19937_block(seed & 0xFFFFFFFF, _p(out), n)
    return out


def philox(k0, k1, c0, c1=0, c2=0, c3=0):
    out = np.zeros(4, np.uint32)
    lib().or_philox4x32_10(k0, k1, c0, c1, c2, c3, _p(out))
    return out


def episode_seed(game, episode):
    """seed16 = (12345 + 7919 i + 104729 e) mod 65536, as int16."""
    v = (12345 + 7919 * np.asarray(game, dtype=np.int64) + 104729 * np.asarray(episode, dtype=np.int64)) & 0xFFFF
    return v.astype(np.uint16).view(np.int16) if isinstance(v, np.ndarray) and v.ndim else np.int16(np.uint16(v).view(np.int16))


# ---------------------------------------------------------------- compiled reference (oracle/_ref)

_ref = None


def ref_available():
    return bool(glob.glob(os.path.join(HERE, "_ref", "tetris_env*.so")))


def ref_module():
    """The compiled reference `tetris_env` (PythonHandle.h:113-340) with a settable clock.

    Returns (module, set_time).  Built by oracle/Makefile from /root/reference; on the GPU box
    only the prebuilt file exists.
    """
    global _ref
    if _ref is None:
        so = glob.glob(os.path.join(HERE, "_ref", "tetris_env*.so"))
        if not so:
            raise FileNotFoundError("oracle/_ref not built (make -C oracle ref; needs /root/reference)")
        spec = importlib.util.spec_from_file_location("tetris_env", so[0])
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        clib = C.CDLL(so[0])
        clib.oracle_set_time.argtypes = [C.c_longlong]
        _ref = (mod, clib.oracle_set_time)
    return _ref
