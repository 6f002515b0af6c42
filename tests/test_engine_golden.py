"""The bitboard engine against the compiled reference's golden traces (tests/golden, made by
make_golden.py from oracle/_ref): every event, every field the reference can show (cells compared
as occupancy).  `harness` = same kernel bodies built by g++ (CPU suite); `hip` = the product on GPU."""
import pytest

from tests import engines, replay

FIELDS = replay.VISIBLE + replay.HIDDEN


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
@pytest.mark.parametrize("name", replay.trace_names())
def test_engine_replays_reference_trace(kind, name):
    trace = replay.load_trace(name)
    max_events = None if kind == "harness" or name in ("greedy_2p", "keys_2p", "greedy_1p", "drop_2p", "rt_2p_sz", "garbage_flood_2p_12") else 700

    def factory(P, H, W, pieces, seed):
        return engines.make(kind, 1, P, H, pieces, seeds=seed)

    n = replay.replay(trace, factory, fields=FIELDS, occupancy_only=True, max_events=max_events, check_actions=True)
    assert n > 0


@pytest.mark.parametrize("kind", engines.ENGINE_PARAMS)
@pytest.mark.parametrize("name", ["greedy_2p", "keys_2p_22", "greedy_2p_o", "rt_2p", "garbage_flood_2p"])
def test_colour_planes_give_the_reference_field_values(kind, name):
    """TETRIS_FLAG_COLOURS: State.field with tile values 1..7 / garbage 8 and GameplayData.garbageCleared exactly as the
    compiled reference shows them (gamePlay.cpp:146,202; gameField.cpp:120-145), over traces with garbage traffic."""
    trace = replay.load_trace(name)

    def factory(P, H, W, pieces, seed):
        return engines.make(kind, 1, P, H, pieces, seeds=seed, colours=True)

    n = replay.replay(trace, factory, fields=FIELDS + replay.COLOUR_ONLY, occupancy_only=False,
                      max_events=None if kind == "harness" else 1200)
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
