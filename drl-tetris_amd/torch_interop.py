"""Zero-copy use of a TetrisBatch from PyTorch-ROCm: actions come from device tensors, observations are written into
device tensors, nothing crosses PCIe.  This is the shape an agent's rollout loop has when its network runs on the same
GPU (worker.py:91-118 with the NN forward pass between get_state and perform_action)."""
import ctypes as C

import numpy as np


class TorchEnv:
    def __init__(self, batch, device=None):
        import torch

        self.torch, self.b = torch, batch
        self.dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        n, P, H, W = batch.n_games, batch.n_players, batch.height, batch.width
        u8 = dict(dtype=torch.uint8, device=self.dev)
        self.done = torch.zeros(n, **u8)
        self.lines = torch.zeros(P, n, **u8)          # player-major on the device (tetris_hip.h)
        self.dead = torch.zeros(P, n, **u8)
        self.visual = torch.zeros(P, n, H, W, **u8)
        self.vector = torch.zeros(P, n, 12, **u8)
        self.piece = torch.zeros(P, n, **u8)
        # run the batch on torch's current stream: kernels are ordered with the surrounding torch ops
        batch.set_stream(torch.cuda.current_stream(self.dev).cuda_stream, external=True)

    def _ptr(self, t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def step_rt(self, rot, trans, player=None, ms=400, auto_reset=False):
        """rot/trans/player: uint8 device tensors [n].  -> (done [n], lines [P,n], dead [P,n]) device tensors (reused).
        auto_reset=True: games whose round ended are reset inside the same launch (drl_tetris/worker.py:157-166 without the
        host round trip; seed schedule of include/tetris_hip.h); the outputs describe the step BEFORE the reset.  Nothing in
        this call waits for the GPU: a rollout loop of observe -> policy -> step_rt(auto_reset=True) runs without a host sync."""
        for t in (rot, trans) + (() if player is None else (player,)):
            assert t.dtype == self.torch.uint8 and t.is_cuda and t.is_contiguous() and t.numel() == self.b.n_games
        self.b.step_rt_dev(self._ptr(rot), self._ptr(trans), self._ptr(player), self._ptr(self.done), self._ptr(self.lines),
                           self._ptr(self.dead), ms=ms, auto_reset=auto_reset)
        return self.done, self.lines, self.dead

    def step_rt_observe(self, rot, trans, player=None, next_player=None, ms=400, auto_reset=False):
        """One iteration of the agent loop in one launch: step_rt(...) and observe(next_player) of the stepped state.
        -> (done, lines, dead, visual, vector, piece) device tensors (reused); bit-identical to the two calls."""
        for t in (rot, trans) + tuple(x for x in (player, next_player) if x is not None):
            assert t.dtype == self.torch.uint8 and t.is_cuda and t.is_contiguous() and t.numel() == self.b.n_games
        self.b.step_rt_observe_dev(self._ptr(rot), self._ptr(trans), self._ptr(player), self._ptr(self.done), self._ptr(self.lines),
                                   self._ptr(self.dead), self._ptr(next_player), self._ptr(self.visual), self._ptr(self.vector),
                                   self._ptr(self.piece), ms=ms, auto_reset=auto_reset)
        return self.done, self.lines, self.dead, self.visual, self.vector, self.piece

    def reset(self, mask=None, seeds=None):
        """Device-side reset: mask uint8 [n] device tensor (non-zero = reset; None = all), seeds int16 [n] device tensor
        (None = the built-in schedule).  Only enqueues."""
        if mask is not None:
            assert mask.dtype == self.torch.uint8 and mask.is_cuda and mask.is_contiguous() and mask.numel() == self.b.n_games
        if seeds is not None:
            assert seeds.dtype == self.torch.int16 and seeds.is_cuda and seeds.is_contiguous() and seeds.numel() == self.b.n_games
        self.b.reset_dev(self._ptr(mask), self._ptr(seeds))

    def observe(self, player=None):
        """-> visual [S,n,H,W], vector [S,n,12], piece [S,n] uint8 device tensors; slot 0 = `player`'s own board."""
        self.b._check(self.b.lib.tetris_observe_packed_dev(self.b._h, None, self.b.n_games, self._ptr(player), self._ptr(self.visual),
                                                          self._ptr(self.vector), self._ptr(self.piece)))
        return self.visual, self.vector, self.piece
