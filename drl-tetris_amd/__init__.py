"""MI355X-native batched Tetris environment — drop-in for DRL-Tetris' environment-step path.

Only what the hot path needs lives here: ``csrc/`` (HIP kernels + C ABI, built into
``lib/libtetris_hip.so``), ``capi`` (ctypes shim) and ``environment`` (the Python host mirror of
the reference's ``tetris_environment_vector`` API).  The directory name contains a hyphen; import
it with ``importlib.import_module("drl-tetris_amd")`` or through ``__graft_entry__.package()``.
"""
from .capi import RECORD, TetrisBatch, TetrisError, load_library  # noqa: F401

__all__ = ["RECORD", "TetrisBatch", "TetrisError", "load_library"]
