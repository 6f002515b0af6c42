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


def test_the_library_finds_its_own_gfx950_code_objects():
    """Direct dispatch (include/tetris_hip.h: tetris_set_direct_dispatch) loads the kernels through the HSA loader from the code
    objects inside the library's own fat binary; if a toolchain change hid them (a compressed offload bundle, say), long rollout
    calls would silently go back to the streams.  Two translation units -> two gfx950 code objects, found without a GPU."""
    import ctypes as C
    lib = ge.package().load_library()
    count, size = C.c_int(-1), C.c_uint64(0)
    assert lib.tetris_debug_code_objects(C.byref(count), C.byref(size)) == 0
    assert count.value == 2 and size.value > 1 << 20


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
