"""The drop-in `tetris_environment_vector` (drl-tetris_amd/environment.py) driven exactly like the reference's
worker loop (drl_tetris/worker.py:91-118) and checked against the oracle stepped per env with the same
seeds/actions: rewards, dones, state_dict observations, reset/set/copy/simulate semantics."""
import numpy as np
import pytest

import __graft_entry__ as ge
from oracle import oracle as orc
from tests import engines

GRID_VALUE_TO_PIECE = {1: 5, 2: 4, 3: 1, 4: 0, 5: 2, 6: 6, 7: 3}     # state_processors.py:24


def _make_env(kind, n, settings):
    pkg = ge.package()
    env_mod = __import__("importlib").import_module("drl-tetris_amd.environment")
    lib = ge.build_harness() if kind == "harness" else None
    return pkg, env_mod, env_mod.tetris_environment_vector(n, None, settings=settings, _lib_path=lib)


class _Clock:
    def __init__(self, t=1000):
        self.t = t

    def __call__(self):
        self.t += 1
        return self.t


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
def test_worker_loop_matches_oracle(kind):
    n, P, H = 24, 2, 22
    clock = _Clock()
    settings = {"n_players": P, "game_size": [H, 10], "seed_source": clock}
    pkg, env_mod, env = _make_env(kind, n, settings)
    edt = __import__("importlib").import_module("drl-tetris_amd.data_types")
    # oracle mirror: construction at seed 1001, then the always-once reset at 1002
    ref = orc.OracleBatch(n, P, H, 10, seeds=1001)
    ref.reset(seeds=1002)
    rng = np.random.default_rng(0)
    current_player = np.zeros(n, np.int64)
    for it in range(120):
        current_player = 1 - current_player                       # worker.py:96
        state = env.get_state()
        assert len(state) == n
        # observation of player p through the lazy processor (state.py:19-27 + state_processors.py:23-54)
        rec = ref.observe()[0]
        for i in (0, n // 2, n - 1):
            for p in range(P):
                d = state[i][p]
                r = rec[i, p]
                assert np.array_equal(d["field"], (r["field"][:H] > 0).astype(np.uint8))
                assert d["field"].dtype == np.uint8 and d["field"].shape == (H, 10)
                cur = GRID_VALUE_TO_PIECE[int(r["grid"].max())]
                assert d["piece"].tolist() == [int(k == cur) for k in range(7)] and d["piece_idx"] == cur
                assert d["nextpiece"].tolist() == [int(k == r["next"]) for k in range(7)]
                assert int(d["x"][0]) == int(np.uint8(r["x"])) and int(d["y"][0]) == int(r["y"])
                assert int(d["incoming_lines"][0]) == int(r["inc_count"])
                assert int(np.ravel(d["combo_time"])[0]) == min(25000, int(r["combo_remaining"]) + 50) // 100 or it == 0
                assert int(d["combo_count"][0]) == int(r["combo_count"])
        rs = rng.integers(0, 4, n)
        ts = rng.integers(0, 10, n)
        actions = [edt.action([8] * int(r) + [2] + [3] * int(t) + [7]) for r, t in zip(rs, ts)]   # sventon_utils.py:9-13
        reward, done = env.perform_action(actions, player=current_player)
        d_ref = ref.step_rt(rs.astype(np.uint8), ts.astype(np.uint8), current_player.astype(np.uint8))
        assert [bool(x) for x in done] == [bool(x) for x in d_ref]
        rec = ref.observe()[0]
        for i in range(n):
            me, you = int(rec[i, current_player[i]]["dead"]), int(rec[i, 1 - current_player[i]]["dead"])
            want = 0 if not d_ref[i] else (-1 if (me and you) else you - me)      # tetris_environment.py:135-149
            assert float(reward[i]()) == float(want)
            assert isinstance(reward[i], edt.maingoal_reward)
        reset_list = [i for i, d in enumerate(done) if d]                          # worker.py:157-160
        env.reset(env=reset_list)
        if reset_list:
            ref.reset(np.array(reset_list, np.int32), seeds=clock.t)
    a = env.backend.observe()[0]
    b = ref.observe()[0]
    for f in ("field", "x", "y", "next", "dead", "time_ms", "piece_draws"):
        fa, fb = (a[f] > 0, b[f] > 0) if f == "field" else (a[f], b[f])
        assert np.array_equal(fa, fb), f


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
def test_set_copy_simulate_and_winner(kind):
    n, P = 6, 2
    pkg, env_mod, env = _make_env(kind, n, {"n_players": P, "game_size": [20, 10], "seed_source": _Clock(50)})
    edt = __import__("importlib").import_module("drl-tetris_amd.data_types")
    drop = edt.action([7])
    for it in range(6):
        env.perform_action([drop] * n, player=it % 2)
    anchor = env.get_state()
    before = env.backend.snapshot()
    # simulate_actions leaves the env untouched and returns one state per action (tetris_environment.py:87-100)
    lists = [edt.action_list([[2, 7], [4, 7], [8, 7]]) for _ in range(n)]
    sims = env.simulate_actions(lists, player=0)
    assert [len(s) for s in sims] == [4] * n            # the null action is put first (action_list.py:8-11)
    assert np.array_equal(env.backend.snapshot(), before)
    fields = [s.backend_state.states[0].field for s in sims[0]]
    assert not np.array_equal(fields[1], fields[2])     # left-most vs right-most drop differ
    # without finalize the piece is stamped but no new piece is dealt (simulate=True, finalize=False)
    sims_nf = env.simulate_actions(lists, player=0, finalize=False)
    assert int(sims_nf[0][1].backend_state.records[0]["time_ms"]) == int(anchor[0].backend_state.records[0]["time_ms"])
    # set() restores a state exactly, RNG position included: replaying gives the same continuation
    env.perform_action([drop] * n, player=0)
    after_a = env.backend.snapshot()
    env.set(anchor)
    assert np.array_equal(env.backend.snapshot(), before)
    env.perform_action([drop] * n, player=0)
    assert np.array_equal(env.backend.snapshot(), after_a)
    # a locked state cannot be changed by actions (state.py:9-12)
    locked = env.get_state()
    for s in locked:
        s.lock()
    env.set(locked)
    frozen = env.backend.observe()[0]["field"].copy()
    env.perform_action([drop] * n, player=0)
    assert np.array_equal(env.backend.observe()[0]["field"], frozen)
    for s in locked:
        s.unlock()
    env.set(locked)
    # copy() gives an independent env in the same state
    twin = env.copy()
    assert np.array_equal(twin.backend.snapshot(), env.backend.snapshot())
    twin.perform_action([drop] * n, player=1)
    assert not np.array_equal(twin.backend.snapshot(), env.backend.snapshot())
    # play player 0 to death: winner is player 1, reward -1 for the loser / +1 seen from the winner
    done = [False] * n
    for it in range(60):
        r, done = env.perform_action([drop] * n, player=0)
        if all(done):
            break
    assert all(done)
    assert env.get_winner() == [1] * n
    assert all(float(x()) == -1.0 for x in r)
    r2, _ = env.perform_action([drop] * n, player=1)         # round over: nothing moves, reward from player 1's view
    assert all(float(x()) == 1.0 for x in r2)
    env.reset(env=[0, 2])
    assert env.get_winner()[:3] == [None, 1, None]


def test_action_list_and_rewards_follow_the_reference_types():
    edt = __import__("importlib").import_module("drl-tetris_amd.data_types")
    al = edt.action_list([[1, 7], [1, 7], [0], [3, 7]], remove_null=True)
    assert [list(a) for a in al] == [[1, 7], [3, 7]] and len(al) == 2
    al2 = edt.action_list([[1, 7]])
    assert [list(a) for a in al2] == [[0], [1, 7]]
    assert [list(a) for a in edt.action_list(None, remove_null=True)] == [[0]]
    a, b = edt.maingoal_reward([1, 5]), edt.maingoal_reward([2, 7])
    assert (a + b).extrinsic.tolist() == [3, 5] and (a - b).extrinsic.tolist() == [-1, 5]
    assert float(edt.maingoal_reward([-1])()) == -1.0
    with pytest.raises(AssertionError):
        edt.action((1, 2))


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
def test_get_actions_simulate_all_and_random_action(kind):
    """sherlock-style use (agents/sherlock_agent/sherlock_agent.py:94-120): enumerate, simulate all afterstates without
    finalizing, pick one, perform it."""
    n, P = 5, 2
    pkg, env_mod, env = _make_env(kind, n, {"n_players": P, "game_size": [20, 10], "seed_source": _Clock(300)})
    edt = __import__("importlib").import_module("drl-tetris_amd.data_types")
    ref = orc.OracleBatch(n, P, 20, 10, seeds=301)
    ref.reset(seeds=302)
    np.random.seed(0)
    for it in range(8):
        p = it % 2
        al = env.get_actions(player=p)
        assert len(al) == n and all(isinstance(x, edt.action_list) for x in al)
        for i in range(n):
            want = edt.action_list(ref.get_actions(i, p), remove_null=True)        # bar_null_moves default True
            assert [list(a) for a in al[i]] == [list(a) for a in want]
        sims = env.simulate_all_actions(player=p, finalize=False)
        assert [len(s) for s in sims] == [len(a) for a in al]
        # every simulated afterstate differs from the current board by exactly one stamped piece (4 cells)
        cur = env.get_state()
        for i in range(n):
            base = cur[i].backend_state.states[p].field > 0
            for st in sims[i]:
                assert int(((st.backend_state.states[p].field > 0) ^ base).sum()) == 4
        acts = env.get_random_action(player=p)
        env.perform_action(acts, player=p)
        keys = np.zeros((n, P, 48), np.uint8); lens = np.ones((n, P), np.uint8)
        for i, a in enumerate(acts):
            keys[i, p, : len(a)] = a; lens[i, p] = len(a)
        ref.make_actions(keys, lens); ref.finish_actions(400)
    a, b = env.backend.observe()[0], ref.observe()[0]
    assert np.array_equal(a["field"] > 0, b["field"] > 0) and np.array_equal(a["next"], b["next"])


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
def test_single_env_sandbox_api(kind):
    """The single-game class agents use as a sandbox (tetris_environment.py:11-227; sherlock_utils.py:13-20)."""
    env_mod = __import__("importlib").import_module("drl-tetris_amd.environment")
    edt = __import__("importlib").import_module("drl-tetris_amd.data_types")
    lib = ge.build_harness() if kind == "harness" else None
    settings = {"n_players": 2, "game_size": [20, 10], "seed_source": _Clock(700)}
    env = env_mod.tetris_environment(settings=settings, _lib_path=lib)
    sandbox = env_mod.tetris_environment(settings=settings, _lib_path=lib)
    ref = orc.OracleBatch(1, 2, 20, 10, seeds=701)
    ref.reset(seeds=702)
    np.random.seed(3)
    for it in range(10):
        p = it % 2
        state = env.get_state()
        actions = sandbox.get_actions(state, player=p)                       # sets the sandbox to `state`, enumerates
        assert [list(a) for a in actions] == [list(a) for a in edt.action_list(ref.get_actions(0, p), remove_null=True)]
        future = sandbox.simulate_all_actions(state, player=p, finalize=False)
        assert len(future) == len(actions)
        a = actions[np.random.randint(len(actions))]
        reward, done = env.perform_action(a, player=p)
        keys = np.zeros((1, 2, 48), np.uint8); lens = np.ones((1, 2), np.uint8)
        keys[0, p, : len(a)] = a; lens[0, p] = len(a)
        ref.make_actions(keys, lens)
        assert bool(ref.finish_actions(400)[0]) == done and float(reward()) in (-1.0, 0.0, 1.0)
    got, want = env.get_state().backend_state.states, ref.observe()[0][0]
    for p in range(2):
        assert np.array_equal(got[p].field > 0, want[p]["field"][:20] > 0) and int(got[p].nextpiece[0]) == int(want[p]["next"])
    twin = env.copy()
    twin.perform_action(edt.action([7]), player=0)
    assert not np.array_equal(twin.backend.snapshot(), env.backend.snapshot())
    assert env.get_winner() is None and "game_size" in str(env)


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
def test_pickle_round_trip_of_env_and_states(kind):
    """tetris_environment_vector.__getstate__/__setstate__ (tetris_environment_vector.py:179-191) and `state` objects
    (workers ship them to the trainer): a pickled env continues exactly like the original, RNG positions included
    (_Clock is a picklable seed source, so later resets agree as well)."""
    import pickle
    settings = {"game_size": [20, 10], "n_players": 2, "pieces": [0, 1, 2, 3, 4, 5, 6], "seed_source": _Clock(50)}
    _, env_mod, env = _make_env(kind, 6, settings)
    rng = np.random.default_rng(3)
    for s in range(12):
        acts = env.get_actions(player=s % 2)
        env.perform_action([a[int(rng.integers(len(a)))] for a in acts], player=s % 2)
    states = env.get_state()
    twin = pickle.loads(pickle.dumps(env))
    again = [pickle.loads(pickle.dumps(st)) for st in states]
    assert all(np.array_equal(a.backend_state.blob, b.backend_state.blob) for a, b in zip(again, states))
    assert np.array_equal(twin.backend.snapshot(), env.backend.snapshot())
    for s in range(60):
        p = s % 2
        a1, a2 = env.get_actions(player=p), twin.get_actions(player=p)
        assert [list(map(list, x)) for x in a1] == [list(map(list, x)) for x in a2]
        pick = [int(rng.integers(len(a))) for a in a1]
        r1, d1 = env.perform_action([a[k] for a, k in zip(a1, pick)], player=p)
        r2, d2 = twin.perform_action([a[k] for a, k in zip(a2, pick)], player=p)
        assert list(d1) == list(d2) and [float(x()) for x in r1] == [float(x()) for x in r2]
        dead = [i for i, d in enumerate(d1) if d]
        env.reset(env=dead)
        twin.reset(env=dead)
    assert np.array_equal(twin.backend.snapshot(), env.backend.snapshot())
    # a restored state sets an env like the original state does
    env.set(again[0], env=[0])
    twin.set(states[0], env=[0])
    assert np.array_equal(twin.backend.snapshot(np.array([0], np.int32)), env.backend.snapshot(np.array([0], np.int32)))


def test_lock_and_set_with_colour_planes_two_players():
    """state.lock() / unlock() + set() on a colour-tracking 2-player env (69 words per player-board, not 39): the dead bit
    Python writes (state.py:9-16) must land in the right player's piece word and nowhere else."""
    n, P = 3, 2
    pkg, env_mod, env = _make_env("harness", n, {"n_players": P, "game_size": [20, 10], "seed_source": _Clock(70), "field_colours": True})
    edt = __import__("importlib").import_module("drl-tetris_amd.data_types")
    for it in range(8):
        env.perform_action([edt.action([7])] * n, player=it % 2)
    before = env.backend.snapshot()
    fields = env.get_fields()
    assert max(int(f.max()) for per_env in fields for f in per_env) > 1          # tile values, not just occupancy
    states = env.get_state()
    for s in states:
        s.lock()
    env.set(states)
    rec = env.backend.observe()[0]
    assert rec["dead"].tolist() == [[1, 1]] * n
    assert all(np.array_equal(a, b) for pa, pb in zip(env.get_fields(), fields) for a, b in zip(pa, pb))   # no colour bit was touched
    for s in states:
        s.unlock()
    env.set(states)
    assert np.array_equal(env.backend.snapshot(), before)
    # one player only
    states[0].backend_state.states[1].dead[0] = 1
    env.set(states)
    assert env.backend.observe()[0]["dead"].tolist() == [[0, 1], [0, 0], [0, 0]]


def test_extra_rewards_two_components():
    """settings['extra_rewards'] (tetris_environment.py:144-149): reward = [w_base * base, w_combo * combo_count] every step."""
    n, P = 8, 2
    pkg, env_mod, env = _make_env("harness", n, {"n_players": P, "game_size": [20, 10], "seed_source": _Clock(5), "pieces": [6],
                                               "extra_rewards": True, "reward_ammount": (1.0, 0.25)})
    edt = __import__("importlib").import_module("drl-tetris_amd.data_types")
    seen_combo = False
    for it in range(60):
        k = (it // 2) % 5
        acts = [edt.action([2] + [3] * (2 * k) + [7]) for _ in range(n)]
        reward, done = env.perform_action(acts, player=it % 2)
        rec = env.backend.observe()[0]
        for i in range(n):
            ext = reward[i].extrinsic
            assert ext.shape == (2,)
            assert ext[1] == 0.25 * int(rec[i, it % 2]["combo_count"])
            seen_combo |= ext[1] > 0
        env.reset(env=[i for i, d in enumerate(done) if d])
    assert seen_combo


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
@pytest.mark.parametrize("P,colours", [(1, False), (2, False), (2, True)])
def test_state_views_decoded_from_snapshot_words_equal_the_observe_records(kind, P, colours):
    """get_state hands out views decoded from the snapshot words (data_types.snapshot_batch); they must be exactly what the
    record kernel (PythonHandle.h:54-82 State views) shows for the same games: field (occupancy, or tile values with colour
    planes), piece grid, x, y, inc_lines, combo_time, combo_count, nextpiece, reward, dead — after play that fills boards, sends
    garbage and ends rounds."""
    n, H = 96, 20
    pkg, env_mod, env = _make_env(kind, n, {"n_players": P, "game_size": [H, 10], "seed_source": _Clock(), "field_colours": colours})
    edt = __import__("importlib").import_module("drl-tetris_amd.data_types")
    rng = np.random.default_rng(3)
    for it in range(90):
        acts = edt.action_batch.from_rt(rng.integers(0, 4, n), (np.arange(n) * 3 + it * 2 + rng.integers(0, 3, n)) % 10)
        _, done = env.perform_action(acts, player=it % P)
        if it % 15 == 14:
            states = env.get_state()
            rec, ro, lw = env.backend.observe()
            for j in range(n):
                b = states[j].backend_state
                assert b.round_over == int(ro[j]) and b.last_winner == int(lw[j])
                for p in range(P):
                    v, r = b.states[p], rec[j, p]
                    assert np.array_equal(v.field, r["field"][:H]) and v.field.dtype == np.uint8
                    assert np.array_equal(v.piece, r["grid"])
                    for name, f in (("x", "x"), ("y", "y"), ("inc_lines", "inc_count"), ("combo_time", "combo_remaining"), ("combo_count", "combo_count"),
                                    ("nextpiece", "next"), ("reward", "reward"), ("dead", "dead")):
                        got = getattr(v, name)
                        assert got.shape == (1,) and got.dtype == r[f].dtype and got[0] == r[f], (name, j, p)
                assert int(b.records[0]["time_ms"]) == int(rec[j, 0]["time_ms"])
        env.reset(env=[i for i, d in enumerate(done) if d])
    assert (rec["field"][:, :, :H] > 1).any() == colours


def test_lazy_list_is_a_list_to_everything_that_looks():
    edt = __import__("importlib").import_module("drl-tetris_amd.data_types")
    import pickle
    made = []
    mk = lambda: edt.lazy_list(5, lambda j: made.append(j) or j * j)
    a = mk()
    assert isinstance(a, list) and len(a) == 5 and bool(a) and not made          # nothing made yet
    assert a[2] == 4 and made == [0, 1, 2, 3, 4] and list(a) == [0, 1, 4, 9, 16] and a[-1] == 16 and a[1:3] == [1, 4]
    assert [7] + mk() == [7, 0, 1, 4, 9, 16] and mk() + [7] == [0, 1, 4, 9, 16, 7]
    assert list(mk()) == [0, 1, 4, 9, 16] and tuple(mk()) == (0, 1, 4, 9, 16) and sorted(mk(), reverse=True)[0] == 16
    assert mk() == [0, 1, 4, 9, 16] and 9 in mk() and mk().index(9) == 3 and [x for x in mk()] == [0, 1, 4, 9, 16]
    assert pickle.loads(pickle.dumps(mk())) == [0, 1, 4, 9, 16]
    b = mk()
    b.append(25)
    assert len(b) == 6 and b[5] == 25
    c = []
    c.extend(mk())
    assert c == [0, 1, 4, 9, 16] and np.asarray(mk()).tolist() == c and [*mk()] == c and list(zip(mk(), range(5)))[4] == (16, 4)
    ab = edt.action_batch.from_rt([0, 3, 1], [0, 9, 4])
    assert len(ab) == 3 and ab.lens.tolist() == [2, 14, 7]
    assert ab[0] == [2, 7] and ab[1] == [8, 8, 8, 2] + [3] * 9 + [7] and ab[2] == [8, 2, 3, 3, 3, 3, 7] and type(ab[0]) is edt.action


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
def test_action_batch_and_action_lists_step_alike_and_untouched_states_restore(kind):
    n, P = 64, 2
    edt = __import__("importlib").import_module("drl-tetris_amd.data_types")
    envs = [_make_env(kind, n, {"n_players": P, "game_size": [20, 10], "seed_source": _Clock()})[2] for _ in range(2)]
    rng = np.random.default_rng(5)
    for it in range(60):
        rot, trans = rng.integers(0, 4, n), rng.integers(0, 10, n)
        r0, d0 = envs[0].perform_action(edt.action_batch.from_rt(rot, trans), player=it % P)
        r1, d1 = envs[1].perform_action([edt.action([8] * int(r) + [2] + [3] * int(t) + [7]) for r, t in zip(rot, trans)], player=it % P)
        assert d0 == d1 and [x() for x in r0] == [x() for x in r1]
        done = [i for i, d in enumerate(d0) if d]
        for e in envs:
            e.reset(env=done)
    assert np.array_equal(envs[0].backend.snapshot(), envs[1].backend.snapshot())
    # states nobody looked at go back into an env as the words they are; looked-at ones through their objects: same result
    anchor = envs[0].get_state()
    envs[0].perform_action(edt.action_batch.from_rt(np.zeros(n, int), np.zeros(n, int)), player=0)
    envs[0].set(anchor)
    assert np.array_equal(envs[0].backend.snapshot(), envs[1].backend.snapshot())
    looked = envs[1].get_state()
    assert len(looked[3]) == P
    envs[0].perform_action(edt.action_batch.from_rt(np.zeros(n, int), np.zeros(n, int)), player=1)
    envs[0].set(looked)
    assert np.array_equal(envs[0].backend.snapshot(), envs[1].backend.snapshot())


def test_get_actions_with_masks_is_one_flag_per_list():
    """PythonHandle.masks[p].mask after get_actions is a vector of ones as long as masks[p].action (TestField.cpp:113-133; probed
    against the compiled reference, 200 calls)."""
    b = engines.make("harness", 4, 2, seeds=np.arange(4))
    lists, masks = b.get_actions(player=1, with_masks=True)
    assert [len(m) for m in masks] == [len(l) for l in lists] and all(set(m) == {1} for m in masks)
    one, m1 = b.get_actions(2, player=0, with_masks=True)
    assert m1 == [1] * len(one) and len(one) > 5


def test_c_packer_and_python_packer_agree():
    """drl-tetris_amd/_fastpack (csrc/fastpack.c) against the pure-Python packer of environment._pack: same arrays, same errors."""
    env_mod = __import__("importlib").import_module("drl-tetris_amd.environment")
    edt = __import__("importlib").import_module("drl-tetris_amd.data_types")
    assert env_mod._fastpack is not None, "build it: python -c 'import __graft_entry__ as g; g.build()'"
    pkg, _, env = _make_env("harness", 300, {"n_players": 2, "game_size": [20, 10], "seed_source": _Clock()})
    rng = np.random.default_rng(9)
    acts = [edt.action(rng.integers(0, 11, int(rng.integers(0, 40))).tolist()) for _ in range(300)]
    who = rng.integers(0, 2, 300)
    fast = env._pack(acts, who, 300)
    keep, env_mod._fastpack = env_mod._fastpack, None
    try:
        slow = env._pack(acts, who, 300)
    finally:
        env_mod._fastpack = keep
    assert np.array_equal(fast[0], slow[0]) and np.array_equal(fast[1], slow[1]) and fast[0].shape == slow[0].shape
    with pytest.raises(TypeError):
        env._pack(acts[:-1] + [[7]], who, 300)             # a plain list is not an `action`
    with pytest.raises(ValueError):
        env._pack(acts[:-1] + [edt.action([256])], who, 300)
