"""C-ABI checks that need no GPU: the HIP library loads, exports every symbol include/*.h declares,
and fails loudly (no CPU fallback) when asked to create a batch without a device."""
import os
import re

import numpy as np
import pytest

import __graft_entry__ as ge


def _declared():
    names = set()
    for fn in os.listdir(os.path.join(ge.ROOT, "include")):
        text = open(os.path.join(ge.ROOT, "include", fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(tetris_[a-z_0-9]+)\s*\(", text))
    return names


def test_library_builds_and_exports_every_declared_symbol():
    ge.build_hip()
    pkg = ge.package()
    lib = pkg.load_library()
    declared = _declared()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"libtetris_hip.so does not export {name}"
    assert set(pkg.capi.EXPORTS) == declared
    assert lib.tetris_record_size() == pkg.RECORD.itemsize
    from oracle import oracle as orc
    assert orc.RECORD == pkg.RECORD            # checker and product agree on the record layout


def test_no_cpu_fallback_without_device():
    pkg = ge.package()
    if ge.gpu_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pkg.TetrisError, match="(?i)hip|device"):
        pkg.TetrisBatch(4, 1)


def test_argument_validation_through_the_harness():
    pkg = ge.package()
    h = ge.build_harness()
    for kw in (dict(n_players=5), dict(n_players=0), dict(height=40), dict(width=12), dict(pieces=[9])):
        args = dict(n_games=2, n_players=2, height=20, width=10, pieces=[0, 1, 2, 3, 4, 5, 6])
        args.update(kw)
        with pytest.raises(pkg.TetrisError):
            pkg.TetrisBatch(lib_path=h, **args)
    b = pkg.TetrisBatch(4, 2, lib_path=h)
    with pytest.raises(pkg.TetrisError):
        b.reset(idx=np.array([7], np.int32))
    with pytest.raises(ValueError):
        b.step_rt(np.zeros(3, np.uint8), np.zeros(3, np.uint8))
