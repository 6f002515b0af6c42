"""The device-pointer entry points (tetris_step_rt_dev, tetris_observe_packed_dev) driven from torch tensors on the batch's
GPU, on torch's stream: an agent-shaped loop (observe -> choose action on device -> step) without any host copy, checked
against the oracle."""
import numpy as np
import pytest

import __graft_entry__ as ge
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def test_zero_copy_loop_matches_oracle():
    import importlib

    import torch
    pkg = ge.package()
    interop = importlib.import_module("drl-tetris_amd.torch_interop")
    n, P, H = 4096, 2, 20
    seeds = orc.episode_seed(np.arange(n), 0)
    batch = pkg.TetrisBatch(n, P, H, 10, seeds=seeds, device=0)
    ref = orc.OracleBatch(n, P, H, 10, seeds=seeds)
    env = interop.TorchEnv(batch)
    gen = torch.Generator(device="cuda").manual_seed(7)
    for s in range(48):
        me = torch.full((n,), s % 2, dtype=torch.uint8, device="cuda")
        visual, vector, piece = env.observe(me)
        # a "policy" computed on the device from the observation: column of the lowest stack + noise, rotation from the piece id
        heights = visual[0].to(torch.int32).flip(1).cumsum(1).gt(0).sum(1)            # [n, W] stack heights of my board
        trans = (heights.argmin(1) + torch.randint(0, 2, (n,), generator=gen, device="cuda")).clamp(0, 9).to(torch.uint8)
        rot = (piece[0] % 4).to(torch.uint8)
        done, lines, dead = env.step_rt(rot, trans, me)
        d_ref = ref.step_rt(rot.cpu().numpy(), trans.cpu().numpy(), me.cpu().numpy())
        assert np.array_equal(done.cpu().numpy(), d_ref), s
        rec = ref.observe()[0]
        assert np.array_equal(dead.cpu().numpy().T, rec["dead"]) and np.array_equal(lines.cpu().numpy().T, rec["reward"])
        idx = np.nonzero(d_ref)[0].astype(np.int32)
        if len(idx):
            torch.cuda.synchronize()
            sd = orc.episode_seed(idx, s + 1)
            batch.reset(idx, sd)
            ref.reset(idx, sd)
    visual, vector, piece = env.observe(torch.zeros(n, dtype=torch.uint8, device="cuda"))
    rec = ref.observe()[0]
    assert np.array_equal(visual[0].cpu().numpy(), (rec["field"][:, 0, :H] > 0).astype(np.uint8))
    assert np.array_equal(visual[1].cpu().numpy(), (rec["field"][:, 1, :H] > 0).astype(np.uint8))
    assert np.array_equal(piece[0].cpu().numpy(), rec["piece"][:, 0])
    batch.close()


def test_out_of_range_device_indices_are_clamped():
    """d_idx / d_player live in HBM and cannot be validated by the host: the kernels clamp them into the batch
    (include/tetris_hip.h), so a bad entry reads the last game / last player instead of faulting."""
    import ctypes as C

    import torch
    pkg = ge.package()
    n, P, H = 512, 2, 20
    batch = pkg.TetrisBatch(n, P, H, 10, seeds=orc.episode_seed(np.arange(n), 0), device=0)
    batch.rollout_random(10, 1)
    ptr = lambda t: C.c_void_p(t.data_ptr())
    m = 64
    bad = torch.tensor([5, -1, 10**6, n, n - 1, 2**31 - 1] + [7] * (m - 6), dtype=torch.int32, device="cuda")
    good = bad.clone().clamp_(0, n - 1)
    good[1] = n - 1                                    # negative reads as a huge unsigned value
    pl_bad = torch.tensor([0, 1, 7, 255] * (m // 4), dtype=torch.uint8, device="cuda")
    pl_good = pl_bad.clamp(max=P - 1)
    outs = []
    for idx, pl in ((bad, pl_bad), (good, pl_good)):
        visual = torch.zeros(P * m * H * 10, dtype=torch.uint8, device="cuda")
        vector = torch.zeros(P * m * 12, dtype=torch.uint8, device="cuda")
        piece = torch.zeros(P * m, dtype=torch.uint8, device="cuda")
        batch._check(batch.lib.tetris_observe_packed_dev(batch._h, ptr(idx), m, ptr(pl), ptr(visual), ptr(vector), ptr(piece)))
        valid = torch.zeros(m * 40, dtype=torch.uint8, device="cuda")
        land = torch.zeros(m * 40, dtype=torch.int8, device="cuda")
        cleared = torch.zeros(m * 40, dtype=torch.uint8, device="cuda")
        batch._check(batch.lib.tetris_enumerate_drops_dev(batch._h, ptr(idx), m, ptr(pl), ptr(valid), ptr(land), ptr(cleared), None))
        batch.sync()
        outs.append([t.cpu().numpy() for t in (visual, vector, piece, valid, land, cleared)])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    batch.close()


def test_zero_copy_loop_with_device_resets_and_no_host_sync():
    """The agent-shaped loop as it runs in production: observe -> policy on the device -> step with TETRIS_STEP_AUTO_RESET, 200
    steps, NOTHING synchronises with the host inside the loop (finished games are reset inside the step launch with the built-in
    seed schedule; worker.py:157-166 without the round trip).  Actions, dones, lines and dead flags are kept as device-side
    histories; afterwards the oracle is driven with the recorded actions and reset like the reference's worker would, and every
    step's outputs plus the final boards are compared."""
    import importlib

    import torch
    pkg = ge.package()
    interop = importlib.import_module("drl-tetris_amd.torch_interop")
    n, P, H, STEPS = 4096, 2, 20, 200
    seeds = orc.episode_seed(np.arange(n), 0)
    batch = pkg.TetrisBatch(n, P, H, 10, seeds=seeds, device=0)
    ref = orc.OracleBatch(n, P, H, 10, seeds=seeds)
    env = interop.TorchEnv(batch)
    gen = torch.Generator(device="cuda").manual_seed(11)
    u8 = dict(dtype=torch.uint8, device="cuda")
    h_rot, h_trans, h_me = torch.zeros(STEPS, n, **u8), torch.zeros(STEPS, n, **u8), torch.zeros(STEPS, n, **u8)
    h_done, h_lines, h_dead = torch.zeros(STEPS, n, **u8), torch.zeros(STEPS, P, n, **u8), torch.zeros(STEPS, P, n, **u8)
    for s in range(STEPS):
        me = torch.full((n,), s % 2, **u8)
        visual, vector, piece = env.observe(me)
        heights = visual[0].to(torch.int32).flip(1).cumsum(1).gt(0).sum(1)            # [n, W] stack heights of my board
        trans = (heights.argmin(1) + torch.randint(0, 3, (n,), generator=gen, device="cuda")).clamp(0, 9).to(torch.uint8)
        rot = ((piece[0] + vector[0][:, 1]) % 4).to(torch.uint8)
        done, lines, dead = env.step_rt(rot, trans, me, auto_reset=True)
        h_rot[s], h_trans[s], h_me[s] = rot, trans, me
        h_done[s], h_lines[s], h_dead[s] = done, lines, dead                         # device-to-device: still no host sync
    torch.cuda.synchronize()
    rots, transs, mes = h_rot.cpu().numpy(), h_trans.cpu().numpy(), h_me.cpu().numpy()
    dones, liness, deads = h_done.cpu().numpy(), h_lines.cpu().numpy(), h_dead.cpu().numpy()
    episode = np.zeros(n, np.int64)
    for s in range(STEPS):
        d_ref = ref.step_rt(rots[s], transs[s], mes[s])
        rec = ref.observe()[0]
        assert np.array_equal(dones[s], d_ref), s
        assert np.array_equal(deads[s].T, rec["dead"]) and np.array_equal(liness[s].T, rec["reward"]), s
        idx = np.nonzero(d_ref)[0].astype(np.int32)
        if len(idx):
            episode[idx] += 1
            ref.reset(idx, orc.episode_seed(idx, episode[idx]))
    assert episode.sum() > n // 2
    from tests import engines
    engines.assert_same_state(batch, ref, where="after 200 un-synchronised steps")
    batch.close()


def test_fused_agent_loop_on_torch_tensors():
    """The agent loop with ONE launch per iteration (TorchEnv.step_rt_observe = tetris_step_rt_observe_dev): the policy reads the
    observation the previous iteration's launch produced, finished games are reset inside the launch, nothing synchronises with
    the host for 200 iterations.  Afterwards the oracle replays the recorded actions; every step's outputs, every step's
    observation (own / opponent planes, vectors, piece ids from the next player's perspective) and the final boards must agree."""
    import importlib

    import torch
    pkg = ge.package()
    interop = importlib.import_module("drl-tetris_amd.torch_interop")
    n, P, H, STEPS = 4096, 2, 20, 200
    seeds = orc.episode_seed(np.arange(n), 0)
    batch = pkg.TetrisBatch(n, P, H, 10, seeds=seeds, device=0)
    ref = orc.OracleBatch(n, P, H, 10, seeds=seeds)
    env = interop.TorchEnv(batch)
    gen = torch.Generator(device="cuda").manual_seed(23)
    u8 = dict(dtype=torch.uint8, device="cuda")
    h_rot, h_trans, h_me = torch.zeros(STEPS, n, **u8), torch.zeros(STEPS, n, **u8), torch.zeros(STEPS, n, **u8)
    h_done, h_lines, h_dead = torch.zeros(STEPS, n, **u8), torch.zeros(STEPS, P, n, **u8), torch.zeros(STEPS, P, n, **u8)
    h_vis, h_vec, h_pc = torch.zeros(STEPS, P, n, H, 10, **u8), torch.zeros(STEPS, P, n, 12, **u8), torch.zeros(STEPS, P, n, **u8)
    me = torch.zeros(n, **u8)
    visual, vector, piece = env.observe(me)                                            # the first decision's observation
    for s in range(STEPS):
        heights = visual[0].to(torch.int32).flip(1).cumsum(1).gt(0).sum(1)            # [n, W] stack heights of my board
        trans = (heights.argmin(1) + torch.randint(0, 3, (n,), generator=gen, device="cuda")).clamp(0, 9).to(torch.uint8)
        rot = ((piece[0] + vector[0][:, 1]) % 4).to(torch.uint8)
        nxt = (1 - me).contiguous()                                                   # worker.py:96: the players alternate
        h_rot[s], h_trans[s], h_me[s] = rot, trans, me
        done, lines, dead, visual, vector, piece = env.step_rt_observe(rot, trans, me, nxt, auto_reset=True)
        h_done[s], h_lines[s], h_dead[s] = done, lines, dead
        h_vis[s], h_vec[s], h_pc[s] = visual, vector, piece
        me = nxt
    torch.cuda.synchronize()
    rots, transs, mes = h_rot.cpu().numpy(), h_trans.cpu().numpy(), h_me.cpu().numpy()
    dones, liness, deads = h_done.cpu().numpy(), h_lines.cpu().numpy(), h_dead.cpu().numpy()
    viss, vecs, pcs = h_vis.cpu().numpy(), h_vec.cpu().numpy(), h_pc.cpu().numpy()
    episode = np.zeros(n, np.int64)
    for s in range(STEPS):
        d_ref = ref.step_rt(rots[s], transs[s], mes[s])
        rec = ref.observe()[0]
        assert np.array_equal(dones[s], d_ref), s
        assert np.array_equal(deads[s].T, rec["dead"]) and np.array_equal(liness[s].T, rec["reward"]), s
        idx = np.nonzero(d_ref)[0].astype(np.int32)
        if len(idx):
            episode[idx] += 1
            ref.reset(idx, orc.episode_seed(idx, episode[idx]))
            rec = ref.observe()[0]
        nxt = 1 - mes[s]
        for sl in range(P):
            r = rec[np.arange(n), nxt if sl == 0 else 1 - nxt]
            assert np.array_equal(viss[s, sl], (r["field"][:, :H] > 0).astype(np.uint8)), (s, sl)
            assert np.array_equal(pcs[s, sl], r["piece"]), (s, sl)
            assert np.array_equal(vecs[s, sl, :, 0], r["x"].astype(np.uint8)) and np.array_equal(vecs[s, sl, :, 2], r["inc_count"]), (s, sl)
    assert episode.sum() > n // 2
    from tests import engines
    engines.assert_same_state(batch, ref, where="after 200 fused iterations")
    batch.close()
