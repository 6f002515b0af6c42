#!/usr/bin/env python3
"""Generates tests/golden/*.npz by RUNNING THE COMPILED REFERENCE (oracle/_ref).

Container-only: needs /root/reference (to build oracle/_ref via `make -C oracle ref`).
The committed .npz files hold data only — inputs (seeds, key sequences) and the outputs the
reference produced (per-step records in the oracle.RECORD layout; fields the reference cannot
show stay zero and are listed in ref_driver.HIDDEN/VISIBLE).  No reference source is stored.

    python tests/golden/make_golden.py            # regenerate everything
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402
from tests.golden.policies import GreedyRT, rt_keys  # noqa: E402
from tests.golden.ref_driver import RefGame  # noqa: E402

MAXK = 32


def record_trace(name, P, H, pieces, seed0, steps, policy, sloppiness=0.0, early_reset_every=0, record_actions=False, solo=None):
    """One game driven for `steps` env-steps; reset with seed0 + 17*episode on done.
    solo = p: only player p ever acts (the others get [0] every step, tetris_environment.py:102-107)."""
    rng = np.random.default_rng(abs(hash((name, seed0))) % (2**32))
    ref = RefGame(P, H, 10, pieces=pieces, seed=seed0)
    shadow = orc.OracleBatch(1, P, H, 10, pieces=pieces, seeds=seed0)   # only feeds the greedy policy
    greedy = GreedyRT(P, H, pieces=pieces, sloppiness=sloppiness, seed=seed0) if policy == "greedy" else None

    ev_kind, ev_seed, ev_player, ev_keys, ev_len, ev_done = [], [], [], [], [], []
    recs, ro, lw = [], [], []
    act_keys, act_lens, act_n, act_player = [], [], [], []     # get_actions(player) BEFORE each event's action
    MAXA, MAXAK = 64, 40

    def snap_actions(player):
        lists = ref.get_actions(player) if record_actions else []
        k = np.zeros((MAXA, MAXAK), np.uint8); l = np.zeros(MAXA, np.uint8)
        assert len(lists) <= MAXA and all(len(a) <= MAXAK for a in lists)
        for i, a in enumerate(lists):
            k[i, : len(a)] = a; l[i] = len(a)
        act_keys.append(k); act_lens.append(l); act_n.append(len(lists)); act_player.append(player)
        return lists

    def snap():
        r, o, w = ref.record()
        recs.append(r); ro.append(o); lw.append(w)

    def do_reset(seed):
        ref.reset(seed); shadow.reset(seeds=seed)
        snap_actions(0)
        ev_kind.append(0); ev_seed.append(seed); ev_player.append(0)
        ev_keys.append(np.zeros(MAXK, np.uint8)); ev_len.append(0); ev_done.append(0)
        snap()

    # event 0: state right after construction (PythonHandle ctor with time()==seed0)
    ev_kind.append(2); ev_seed.append(seed0); ev_player.append(0)
    ev_keys.append(np.zeros(MAXK, np.uint8)); ev_len.append(0); ev_done.append(0)
    snap_actions(0)
    snap()
    do_reset(seed0)     # tetris_environment.__init__ always resets once (tetris_environment.py:40-41)
    episode = 0
    for s in range(steps):
        player = s % P if solo is None else solo
        lists = snap_actions(player)
        if policy == "actions":
            keys = lists[int(rng.integers(len(lists)))]
        elif policy == "rt":
            keys = rt_keys(rng.integers(4), rng.integers(10))
        elif policy == "keys":
            keys = list(rng.integers(0, 11, size=rng.integers(0, 14)))
            if rng.random() < 0.75:
                keys.append(7)
            if rng.random() < 0.05:
                keys += [int(rng.integers(8, 11)), 7]      # keys after a lock
        elif policy == "greedy":
            keys = rt_keys(*greedy.choose(shadow, 0, player))
        elif policy == "drop":
            keys = [7]
        else:
            raise ValueError(policy)
        keys = [int(k) for k in keys][:MAXK]
        done = ref.step(keys, player)
        K = np.zeros((1, P, MAXK), np.uint8); L = np.ones((1, P), np.uint8)
        K[0, player, : len(keys)] = keys; L[0, player] = len(keys)
        shadow.make_actions(K, L); shadow.finish_actions(400)
        kk = np.zeros(MAXK, np.uint8); kk[: len(keys)] = keys
        ev_kind.append(1); ev_seed.append(0); ev_player.append(player)
        ev_keys.append(kk); ev_len.append(len(keys)); ev_done.append(int(done))
        snap()
        if done or (early_reset_every and s % early_reset_every == early_reset_every - 1):
            episode += 1
            do_reset(seed0 + 17 * episode)
    out = dict(
        n_players=P, height=H, width=10, pieces=np.array((list(pieces) * 7)[:7], np.uint8), ms=400,
        ev_kind=np.array(ev_kind, np.uint8),          # 2 = constructed, 0 = reset(seed), 1 = step
        ev_seed=np.array(ev_seed, np.int64), ev_player=np.array(ev_player, np.uint8),
        ev_keys=np.stack(ev_keys), ev_len=np.array(ev_len, np.uint8), ev_done=np.array(ev_done, np.uint8),
        records=np.stack(recs), round_over=np.array(ro, np.uint8), last_winner=np.array(lw, np.int8),
    )
    if record_actions:
        out.update(act_keys=np.stack(act_keys), act_lens=np.stack(act_lens), act_n=np.array(act_n, np.int32),
                   act_player=np.array(act_player, np.uint8))
    path = os.path.join(HERE, f"trace_{name}.npz")
    np.savez_compressed(path, **out)
    cleared = int(out["records"]["reward"][out["ev_kind"] == 1].sum())
    print(f"{name}: {len(ev_kind)} events, {episode} resets, {cleared} lines, {os.path.getsize(path) / 1024:.0f} KiB")


def rotation_table():
    """All 7 pieces: spawn state, then cw x4, ccw x4, 180 x2 on an empty 20x10 board, at spawn and
    pushed against both walls (kick table, gameField.cpp:55-103)."""
    rows = []
    for piece in range(7):
        for prefix in ([], [2], [4], [2, 3], [4, 1]):
            for seq in ([8] * 4, [9] * 4, [10] * 2, [8, 10, 9], [5, 5, 8, 9, 10]):
                g = RefGame(1, 20, 10, pieces=[piece], seed=3)
                g.reset(3)
                for k in prefix:
                    g.make([[k]])
                for k in seq:
                    g.make([[k]])
                    r, _, _ = g.record()
                    rows.append((piece, len(prefix) and prefix[0], k, int(r["x"][0]), int(r["y"][0]), int(r["cur_rot"][0]),
                                 r["grid"][0].copy()))
    arr = np.zeros(len(rows), dtype=[("piece", "u1"), ("prefix", "u1"), ("key", "u1"), ("x", "i1"), ("y", "i1"), ("rot", "u1"), ("grid", "u1", (4, 4))])
    for i, row in enumerate(rows):
        arr[i] = row
    return arr


def rng_kat():
    """First pieces per seed for several piece sets (randomizer.cpp:10-62, gamePlay.cpp:218-230)."""
    out = {}
    for tag, pieces in (("all", [0, 1, 2, 3, 4, 5, 6]), ("sz", [2, 3]), ("i", [4]), ("lj", [0, 1])):
        seeds = [1000, 0, 1, -1, -5, 12345, 32767, -32768, 40000, 66536]
        seq = np.zeros((len(seeds), 40), np.uint8)
        for i, sd in enumerate(seeds):
            g = RefGame(1, 20, 10, pieces=pieces, seed=sd)
            g.reset(sd)
            r, _, _ = g.record()
            cur = {4: 0, 3: 1, 5: 2, 7: 3, 2: 4, 1: 5, 6: 6}[int(r["grid"][0].max())]
            seq[i, 0], seq[i, 1] = cur, r["next"][0]
            for j in range(2, 40):
                g.make([[1]])        # never locks: board stays empty, every finish deals a new piece
                g.finish(400)
                seq[i, j] = g.record()[0]["next"][0]
        out[f"seeds_{tag}"] = np.array(seeds, np.int64)
        out[f"pieces_{tag}"] = seq
    return out


TRACES = {
    "rt_1p": dict(P=1, H=20, pieces="all", seed0=1000, steps=1500, policy="rt"),
    "rt_2p": dict(P=2, H=20, pieces="all", seed0=1000, steps=2500, policy="rt", early_reset_every=333),
    "rt_2p_neg": dict(P=2, H=20, pieces="all", seed0=-5, steps=1200, policy="rt"),
    "keys_1p": dict(P=1, H=20, pieces="all", seed0=40000, steps=1500, policy="keys"),
    "keys_2p": dict(P=2, H=20, pieces="all", seed0=7, steps=2500, policy="keys"),
    "keys_2p_22": dict(P=2, H=22, pieces="all", seed0=31000, steps=1500, policy="keys"),
    "greedy_1p": dict(P=1, H=20, pieces="all", seed0=1000, steps=3000, policy="greedy"),
    "greedy_2p": dict(P=2, H=20, pieces="all", seed0=1000, steps=3000, policy="greedy", sloppiness=0.05),
    "greedy_2p_b": dict(P=2, H=22, pieces="all", seed0=77, steps=2500, policy="greedy", sloppiness=0.15),
    "greedy_1p_io": dict(P=1, H=20, pieces=[4, 6], seed0=9, steps=1500, policy="greedy", sloppiness=0.02),
    "greedy_2p_o": dict(P=2, H=20, pieces=[6], seed0=11, steps=1500, policy="greedy", sloppiness=0.02),
    "rt_2p_sz": dict(P=2, H=20, pieces=[2, 3], seed0=5, steps=800, policy="rt"),
    "drop_2p": dict(P=2, H=20, pieces="all", seed0=1000, steps=120, policy="drop"),
    # get_actions() lists (TestField.cpp:64-415) recorded before every step; the policy plays one of them, so
    # tucks and spins create overhangs that later enumerations have to reach
    "actions_2p": dict(P=2, H=20, pieces="all", seed0=4, steps=500, policy="actions", record_actions=True),
    "actions_1p_22": dict(P=1, H=22, pieces="all", seed0=21, steps=400, policy="actions", record_actions=True),
    "actions_2p_greedy": dict(P=2, H=20, pieces="all", seed0=3, steps=300, policy="greedy", sloppiness=0.3, record_actions=True),
    # player 0 clears and sends, player 1 never acts: its board fills with garbage rows only, until a pushed row cannot
    # lift the freshly dealt piece any more (gamePlay.cpp:179-192 pushGarbage -> death) or the spawn collides
    "garbage_flood_2p": dict(P=2, H=20, pieces=[6, 4], seed0=23, steps=2500, policy="greedy", sloppiness=0.02, solo=0),
    "garbage_flood_2p_12": dict(P=2, H=12, pieces=[6], seed0=8, steps=1500, policy="greedy", sloppiness=0.0, solo=0),
    # more than two players (PythonHandle.cpp:5-25, distributeLines :124-136: every opponent receives amount / (P - 1) lines as a
    # float — halves with three players, thirds with four — and the round ends when fewer than two are alive)
    "rt_3p": dict(P=3, H=20, pieces="all", seed0=1000, steps=2400, policy="rt"),
    "keys_4p": dict(P=4, H=20, pieces="all", seed0=12, steps=2000, policy="keys"),
    "greedy_3p_io": dict(P=3, H=20, pieces=[6, 4], seed0=5, steps=2400, policy="greedy", sloppiness=0.03),
    "greedy_4p_o": dict(P=4, H=20, pieces=[6], seed0=19, steps=2400, policy="greedy", sloppiness=0.03),
}


def main():
    """no arguments: every trace + the tables (NOTE: the random policies are seeded from hash((name, seed)), which
    differs between Python processes unless PYTHONHASHSEED is fixed — regenerate single traces by name instead:
        python tests/golden/make_golden.py garbage_flood_2p ...)"""
    orc.build()
    if not orc.ref_available():
        sys.exit("oracle/_ref is not built and /root/reference is absent: cannot generate fixtures here")
    allp = [0, 1, 2, 3, 4, 5, 6]
    names = sys.argv[1:] or list(TRACES)
    for name in names:
        kw = dict(TRACES[name])
        if kw["pieces"] == "all":
            kw["pieces"] = allp
        record_trace(name, kw.pop("P"), kw.pop("H"), kw.pop("pieces"), kw.pop("seed0"), kw.pop("steps"), kw.pop("policy"), **kw)
    if not sys.argv[1:]:
        np.savez_compressed(os.path.join(HERE, "rotation_table.npz"), table=rotation_table())
        np.savez_compressed(os.path.join(HERE, "rng_kat.npz"), **rng_kat())
    print("done")


if __name__ == "__main__":
    main()
