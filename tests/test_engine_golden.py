"""The bitboard engine against the compiled reference's golden traces (tests/golden, made by
make_golden.py from oracle/_ref): every event, every field the reference can show (cells compared
as occupancy).  `harness` = same kernel bodies built by g++ (CPU suite); `hip` = the product on GPU."""
import pytest

from tests import engines, replay

FIELDS = replay.VISIBLE + replay.HIDDEN


@pytest.mark.parametrize("name", replay.trace_names())
def test_engine_replays_reference_trace(name):
    """Every trace on its own, full length, through a 1-game batch of the CPU harness build (the HIP path replays the same
    traces at full length as the games of one batch per geometry: test_all_traces_full_length_in_one_batch)."""
    trace = replay.load_trace(name)

    def factory(P, H, W, pieces, seed):
        return engines.make("harness", 1, P, H, pieces, seeds=seed)

    n = replay.replay(trace, factory, fields=FIELDS, occupancy_only=True, check_actions=True)
    assert n > 0


GROUPS = sorted(replay.trace_groups().items())


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
@pytest.mark.parametrize("colours", [False, True])
@pytest.mark.parametrize("key,names", GROUPS, ids=["-".join(names) for _, names in GROUPS])
def test_all_traces_full_length_in_one_batch(kind, colours, key, names):
    """All 18 reference traces at FULL length (no event cut): the traces of one geometry are the games of ONE batch, so every
    event index is one reset call + one make/finish pair + one observe for all of them — 30 000 reference events through
    the batched entry points with index lists, `get_actions` lists included where the reference recorded them.  With
    colours: State.field values 0..8 and garbageCleared exactly (gamePlay.cpp:146,202; gameField.cpp:120-145)."""
    def factory(n, P, H, W, pieces, seeds):
        return engines.make(kind, n, P, H, pieces, seeds=seeds, colours=colours)

    pairs = replay.replay_batch(names, factory, fields=FIELDS + (replay.COLOUR_ONLY if colours else []), occupancy_only=not colours,
                                check_actions=not colours)
    assert pairs == sum(len(replay.load_trace(n)["ev_kind"]) for n in names)


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
@pytest.mark.parametrize("name", ["greedy_2p", "keys_2p_22", "greedy_2p_o", "rt_2p", "garbage_flood_2p"])
def test_colour_planes_give_the_reference_field_values(kind, name):
    """TETRIS_FLAG_COLOURS: State.field with tile values 1..7 / garbage 8 and GameplayData.garbageCleared exactly as the
    compiled reference shows them (gamePlay.cpp:146,202; gameField.cpp:120-145), over traces with garbage traffic."""
    trace = replay.load_trace(name)

    def factory(P, H, W, pieces, seed):
        return engines.make(kind, 1, P, H, pieces, seeds=seed, colours=True)

    n = replay.replay(trace, factory, fields=FIELDS + replay.COLOUR_ONLY, occupancy_only=False,
                      max_events=None if kind == "harness" else 300)
    assert n > 0


def test_golden_traces_reach_the_rare_paths():
    """The fixtures are only worth their bytes if replaying them runs the branches they were recorded for: counted in the
    CPU harness build (tetris_engine.h TE_COUNT) over the traces that were made for garbage and lock-down."""
    engines.harness_path_counts()                      # clear
    total = dict.fromkeys(engines.PATH_COUNTERS, 0)
    for name in ("garbage_flood_2p", "garbage_flood_2p_12", "greedy_2p", "keys_2p", "greedy_1p"):
        trace = replay.load_trace(name)
        replay.replay(trace, lambda P, H, W, pieces, seed: engines.make("harness", 1, P, H, pieces, seeds=seed), fields=FIELDS,
                      occupancy_only=True)
        for k, v in engines.harness_path_counts().items():
            total[k] += v
    for name in ("garbage_row", "death_garbage", "death_spawn", "timer_lock", "key_kick", "key_kick_failed"):
        assert total[name] > 0, (name, total)
    # gamePlay.cpp:184-190: the second lift cannot happen — a piece lifted with the stack keeps its position relative to it
    assert total["garbage_lift2"] == 0
