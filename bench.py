#!/usr/bin/env python3
"""Benchmark of the hot path: env-steps/s of seeded random-policy rollouts.

    python bench.py --gpus N --steps K --warmup W

One "step" = one launch of the rollout kernel = ONE env-step on every game of the batch (state is
read from HBM, stepped, auto-reset where done, written back).  N=1 workload = BASELINE.json
configs[1]: 65 536 parallel 20x10 single-player boards, random (rotation, translation) policy.
For N>1 every rank (one process per GPU) owns its own 65 536 games: games never interact across ranks, so there is no
data-path collective ("scaling": "weak").  `python bench.py --gpus N` starts its own N ranks (a parent that touches no GPU
spawns one child per device and relays rank 0's line; fewer than N visible devices, or a rank that fails, is a non-zero
exit — never a silent run on fewer GPUs); under an external launcher (torch.distributed.run: RANK / WORLD_SIZE in the
environment) it is one of the launcher's ranks, and WORLD_SIZE must equal --gpus.

Prints ONE JSON line (rank 0).  `value` = env-steps of all ranks / max-over-ranks wall time of the
K timed launches, inputs resident in HBM; nothing but the K step-kernel launches lies inside that
region (counters are summed by a separate kernel before and after it).  `roofline` prices the step
kernel against the 8 TB/s HBM roofline with SURVEY §8(d)'s algorithmic bytes (389 B per 1-player
env-step, 774 B per 2-player one) over the SAME wall clock as `value`; `launch_us_events` /
`frac_kernel` give the same two figures from HIP events on the launch stream.
`cpu_baseline` times the reference C++ backend itself (oracle/_ref, built in the container from
/root/reference) or, without it, the C restatement, on the host cores of this box.
"""
import argparse
import json
import os
import sys
import time


import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402

ALGO_BYTES = {1: 389, 2: 774}       # P*(192 read + 192 write) + 2 action + (P+2) outputs
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s spec


def cpu_baseline(n_players, height, budget_s):
    """Reference (or port) env-steps/s on this box's host cores, bounded sample of the same workload."""
    from oracle import oracle as orc

    out = {}
    n = 32
    if orc.ref_available():
        # the compiled reference backend driven like tetris_environment.perform_action does
        # (make_action + finish_action, reset when done), single thread, 32 envs round-robin
        mod, set_time = orc.ref_module()
        mod.set_pieces([0, 1, 2, 3, 4, 5, 6])
        envs = []
        for g in range(n):
            set_time(int(orc.episode_seed(g, 0)))
            envs.append(mod.PythonHandle(n_players, [height, 10]))
        episode = [0] * n
        steps = 0
        t0 = time.perf_counter()
        s = 0
        while time.perf_counter() - t0 < budget_s:
            for g, h in enumerate(envs):
                w = orc.philox(0xD71, 0, g, s & 0xFFFFFFFF, s >> 32, 0)
                keys = [8] * int(w[0] & 3) + [2] + [3] * int(w[1] % 10) + [7]
                a = [[0] for _ in range(n_players)]
                a[s % n_players] = keys
                h.make_action(a)
                if h.finish_action(400):
                    episode[g] += 1
                    set_time(int(orc.episode_seed(g, episode[g])))
                    h.reset()
            s += 1
            steps += n
        dt = time.perf_counter() - t0
        out = {"value": steps / dt, "unit": "env-steps/s", "cores": 1, "kind": "reference",
               "sample": f"reference C++ backend (oracle/_ref, -O0 as the reference builds it), {n} envs x {s} steps, "
                         f"{n_players} player(s), {height}x10, same policy/seeds, bare make_action+finish_action+reset loop from Python"}
    # the C restatement (port), one thread and all cores
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cores = min(cores, 32)
    nb = 4096
    b = orc.OracleBatch(nb, n_players, height, 10, seeds=orc.episode_seed(np.arange(nb), 0))
    port = {}
    for threads in (1, cores):
        ep = np.zeros(nb, np.uint32)
        t0 = time.perf_counter()
        done_steps = 0
        first = 0
        while time.perf_counter() - t0 < max(1.0, budget_s / 4):
            b.rollout_random(16, first_step=first, episode=ep, threads=threads)
            first += 16
            done_steps += 16 * nb
        port[threads] = done_steps / (time.perf_counter() - t0)
    if not out:
        out = {"value": port[cores], "unit": "env-steps/s", "cores": cores, "kind": "port",
               "sample": f"oracle C restatement, {nb} envs, OpenMP over envs"}
    out["port_1_thread"] = port[1]
    out["port_all_cores"] = {"value": port[cores], "cores": cores}
    return out


def bench_split(args, mod, dist, rank, world, local_rank):
    """BASELINE config 5: opponents on different GPUs, RCCL all-gather for the garbage exchange (SplitOpponents)."""
    if dist is None or world % 2:
        raise SystemExit("--workload split needs an even number of ranks >= 2 (torch.distributed.run)")
    import torch
    N, pair = args.games, rank // 2
    so = mod.SplitOpponents(N, side=rank % 2, peer=rank ^ 1, dist=dist, height=args.height, seeds=mod.episode_seeds(pair * N, N),
                            device=local_rank, lib_path=os.environ.get("BENCH_LIB_PATH"))
    so.batch.set_game_offset(pair * N)
    so.rollout(args.warmup)
    before = [int(x) for x in so.batch.rollout_totals()]
    dist.barrier()
    wall = so.rollout(args.steps, first_step=args.warmup)
    dist.barrier()
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([wall], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tot = torch.tensor([int(x) - b for x, b in zip(so.batch.rollout_totals(), before)], dtype=torch.int64, device=dev)      # the timed steps only
    if rank % 2:
        tot[0] = 0                                   # env-steps and episodes are counted once per game (by side 0)
        tot[1] = 0
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    if rank == 0:
        wall = float(t[0])
        env_steps = (world // 2) * N * args.steps
        per_step_us = wall * 1e6 / args.steps
        algo = (ALGO_BYTES[2] // 2) * N                # one player-board of every game per GPU and step
        achieved = algo / (per_step_us * 1e-6) / 1e9
        print(json.dumps({
            "metric": "env-steps/sec at 64k parallel 20x10 boards", "value": env_steps / wall, "unit": "env-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{N} two-player {args.height}x10 games per GPU pair, player 0 and player 1 on different GPUs, "
                                   "random policy, auto-reset, garbage exchange = 3 all-gathers of 4 B per board per step",
                       "games_per_pair": N, "players": 2, "parallelism": f"{world // 2} pair(s) of GPUs, every exchange an all-gather inside the pair's own process group (2 ranks)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "k_split<0..2> + 3 all-gathers per step (wall time, not kernel time)",
                         "algorithmic_bytes_per_launch": algo, "launch_us": per_step_us},
            "env_steps_counted_on_device": int(tot[0]), "episodes": int(tot[1]), "lines_cleared": int(tot[2]), "garbage_sent": int(tot[3]),
            "cpu_baseline": None}))
    so.close()
    dist.destroy_process_group()


def visible_devices():
    """GPUs this process may use, WITHOUT initialising the HIP runtime (the parent of the ranks must stay free of it: a
    process that has touched the GPU must not start other programs on this pool): torch.cuda.device_count() reads the
    driver's device list only."""
    import torch
    return int(torch.cuda.device_count())


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: N child processes, one per device, rendezvous on 127.0.0.1; rank 0's
    stdout (the JSON line) is relayed, every rank's stderr passes through.  Returns the exit code."""
    import socket
    import subprocess
    n = args.gpus
    rehearsal = os.environ.get("BENCH_LIB_PATH") is not None          # CPU rehearsal (tests): no devices to count
    if not rehearsal:
        have = visible_devices()
        if have < n:
            print(f"bench.py: --gpus {n} but only {have} GPU(s) visible to this process; refusing to run on fewer", file=sys.stderr)
            return 2
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    deadline = time.time() + float(os.environ.get("BENCH_SPAWN_TIMEOUT_S", "1500"))
    rc, pending = 0, set(range(n))
    while pending and not rc:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is not None:
                pending.discard(r)
                if code != 0:
                    print(f"bench.py: rank {r} exited with code {code}", file=sys.stderr)
                    rc = code if code > 0 else 1
        if time.time() > deadline:
            print("bench.py: ranks did not finish in time", file=sys.stderr)
            rc = 3
        if pending and not rc:
            time.sleep(0.05)
    for r in pending:                                  # a rank failed (or the deadline passed): end the others — these exact processes
        procs[r].kill()
    out = procs[0].stdout.read() if procs[0].stdout else ""
    for pr in procs:
        pr.wait()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if not rc and len(lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(lines)}", file=sys.stderr)
        rc = 4
    if not rc:
        line = json.loads(lines[0])
        if line.get("n_gpus") != n:
            print(f"bench.py: rank 0 reports n_gpus {line.get('n_gpus')} for --gpus {n}", file=sys.stderr)
            rc = 5
        else:
            print(lines[0])
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2048)
    ap.add_argument("--warmup", type=int, default=64)
    ap.add_argument("--games", type=int, default=65536, help="games per GPU")
    ap.add_argument("--players", type=int, default=1)
    ap.add_argument("--height", type=int, default=20)
    ap.add_argument("--steps-per-launch", type=int, default=1, help=">1 = fused rollout (state stays in registers)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--precondition-ms", type=float, default=40.0,
                    help="before the W warm-up steps: this many ms of the same launches on a SCRATCH batch (device clocks take tens of ms "
                         "to leave their idle state; the measured batch, its W warm-up steps and its K timed steps are untouched by it). 0 = off")
    ap.add_argument("--workload", choices=["sharded", "split"], default="sharded",
                    help="sharded (default): every rank owns whole games, no collective.  split: BASELINE config 5 — ranks 2k and 2k+1 "
                         "hold player 0 / player 1 of the same games, garbage exchange by three RCCL all-gathers per step (even N only)")
    ap.add_argument("--no-gpu-paced", action="store_true", help="skip the extra GPU-paced run after the timed region (profiling runs)")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(spawn_ranks(args))                # parent: no GPU call, no HIP library in this process
    elif int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        # under a launcher: its world IS the job.  A mismatch would print a line about a different job than the one asked for.
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')} rank(s)", file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):         # (BENCH_FORCE_DIST=1: a one-rank process group, to rehearse the RCCL calls on a one-GPU box)
        import torch
        import torch.distributed as dist

        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")     # "gloo" only to rehearse N>1 on a 1-GPU box
        if backend == "nccl" and int(os.environ.get("LOCAL_WORLD_SIZE", world)) > torch.cuda.device_count():
            print(f"bench.py: {os.environ.get('LOCAL_WORLD_SIZE', world)} ranks on this node but {torch.cuda.device_count()} GPU(s) visible", file=sys.stderr)
            sys.exit(2)
        local_rank = local_rank % max(1, torch.cuda.device_count())      # (wraps only in the gloo rehearsal of N ranks on fewer devices)
        if torch.cuda.is_available():                              # (absent only in the CPU rehearsal with gloo + BENCH_LIB_PATH)
            torch.cuda.set_device(local_rank)
        # RCCL prints its version banner on STDOUT when the first communicator is made (the first collective); stdout is for the one
        # JSON line, so file descriptor 1 points at stderr until the process group is up and a first barrier has gone through
        sys.stdout.flush()
        keep_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
            dist.barrier()
            if torch.cuda.is_available():
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(keep_fd, 1)
            os.close(keep_fd)
    ge.package()
    import importlib
    sharded = importlib.import_module("drl-tetris_amd.distributed")
    capi = importlib.import_module("drl-tetris_amd.capi")
    if args.workload == "split":
        return bench_split(args, sharded, dist, rank, world, local_rank)
    N, P, S = args.games, args.players, args.steps_per_launch
    # rank r owns global games [r*N, (r+1)*N): distinct policy stream and seed schedule per rank, no collective
    # on the data path; ShardedRollout brackets the timed launches with barrier + synchronize on both sides (each rank's clock
    # stops when ITS launches have drained, before the closing barrier: that barrier is no part of the K steps and over RCCL it
    # costs as much as a short rollout) and reduces time (MAX) and counters (SUM) over the ranks.
    shard = sharded.ShardedRollout(N, P, args.height, 10, rank=rank, world=world, device=local_rank, dist=dist,
                                   lib_path=os.environ.get("BENCH_LIB_PATH"))   # (set only to rehearse the N>1 launch without GPUs)
    precondition_launches = 0
    if args.precondition_ms > 0 and not os.environ.get("BENCH_LIB_PATH"):
        # A freshly started process finds the GPU at idle clocks; they take tens of milliseconds of load to settle (the first
        # 2048-launch run of a process measured 5.96 us per launch, every later one 4.96: profiles/r02/order_check.txt), far
        # longer than W = 5 warm-up launches.  So the device — not the measured batch — is warmed first: the same kind of
        # launches on a scratch batch of the same shape, which is then thrown away.  Untimed, stated in the JSON line.
        scratch = sharded.ShardedRollout(N, P, args.height, 10, rank=rank, world=world, device=local_rank, dist=None)
        # (calls of the timed call's own length, at most 256 launches: the library sends short and long calls down different launch
        # paths — streams / its own queues, include/tetris_hip.h: tetris_set_direct_dispatch — and the path that is timed is the one warmed)
        per_call = max(1, min(256, args.steps))
        t_end = time.perf_counter() + args.precondition_ms * 1e-3
        while time.perf_counter() < t_end:
            scratch.batch.rollout_launch(per_call, S, first_step=precondition_launches * S)
            precondition_launches += per_call
        scratch.close()
    if args.warmup > 0:
        shard.run(args.warmup, S)                     # untimed warm-up
    res = shard.run(args.steps, S)                    # exactly K timed launches
    counters, wall, ev_ms = res["counters"], res["wall_s"], res["event_ms"]
    direct = bool(shard.batch.rollout_was_direct())       # (asked now: the GPU-paced extra run below goes through the streams)
    affine = bool(shard.batch.rollout_was_affine())       # the XCD-affine kernels (include/tetris_hip.h: tetris_set_xcd_affine)
    batch = shard
    # after the timed region, single GPU, chained single-step launches only: the same launches GPU-paced (the library parks its
    # streams behind a blocker kernel until all 512 launches are queued) — the period the GPU sustains when the host's launch
    # cost (2.5-5 us per launch, it varies between processes and boxes and sets the pace where it exceeds the GPU's period)
    # does not enter.  Reported beside the wall-clock figures, never as `value`.
    gpu_paced_us = None
    if world == 1 and S == 1 and not args.no_gpu_paced and shard.batch.rollout_is_chained(S) and os.environ.get("BENCH_LIB_PATH") is None:
        os.environ["TETRIS_PREQUEUE"] = "1"
        try:
            shard.run(64, S)
            gpu_paced_us = min(shard.run(512, S)["event_ms"] for _ in range(2)) * 1e3 / 512
        except Exception as e:                                   # (an extra, never a reason to lose the line)
            print(f"bench: GPU-paced extra run failed: {e}", file=sys.stderr)
            gpu_paced_us = None
        finally:
            del os.environ["TETRIS_PREQUEUE"]
    # chained launches (include/tetris_hip.h: tetris_set_chained): on by default where two launches fit on the device together —
    # 64k single-player boards do, 64k two-player boards (k_duo needs 220 VGPRs) do not
    chained = shard.batch.rollout_is_chained(S)
    if rank == 0:
        # env-steps are COUNTED by the step kernels (one increment per game and step in a per-game word, summed by a separate
        # kernel before and after the timed region): the comparison below is a check of the device, not of the host's arithmetic
        env_steps = int(counters[0])
        assert env_steps == world * N * S * args.steps, ("env-steps counted on the device differ from launches*S*N", env_steps, world, N, S, args.steps)
        # ONE clock for value and roofline.frac: the wall clock around the K launches (barrier + synchronize on both sides).
        # The HIP events around the same K launches on the launch stream are reported beside it (`launch_us_events`,
        # `frac_kernel`); with nothing but the step kernels inside the region the two agree to a few per cent.
        launch_us = wall * 1e6 / args.steps
        launch_us_events = ev_ms * 1e3 / args.steps if ev_ms > 0 else None     # (the CPU rehearsal library has no events)
        lib_path = os.path.abspath(os.environ.get("BENCH_LIB_PATH") or ge.LIB)
        roofline = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "kernel": (((("k_chain_affine<1>" if P == 1 else "k_duo_affine") + " — XCD-affine: a game block is stepped on the same XCD in every launch, its state stays in that XCD's L2 —")
                                if affine else ("k_chain<1>" if P == 1 else "k_duo<M_ROLLOUT, true>")) + " (chained launches: consecutive launches on "
                               + ("three" if P == 1 else "two") + " queues / streams, each wave waits for its own predecessor's epoch word)") if chained
                              else ("k_duo<M_ROLLOUT>" if (P == 2 and S == 1) else f"k_game<{P}, M_ROLLOUT>"),
                    "launch_us": launch_us, "launch_us_events": launch_us_events, "clock": "wall (same clock as `value`)"}
        if chained:
            roofline["dispatch"] = ("direct: the K launches are AQL packets the library writes into HSA queues of its own (three per GPU; include/tetris_hip.h: "
                                    "tetris_set_direct_dispatch); `launch_us_events` = (end of the last dispatch - start of the first) / K from the packets' own timestamps"
                                    if direct else "streams: hipLaunchKernel on the batch's chain streams; `launch_us_events` from HIP events attached to the first and last kernel")
            roofline["launch_us_is"] = ("the launch PERIOD: consecutive launches overlap (a wave of launch E starts as soon as the same wave of "
                                        "launch E-1 has published its state), so the durations in a kernel trace are longer than the period "
                                        "(profiles/r03/final/chain_period_from_trace_p*.json derives the period from the trace's own timestamps); "
                                        "TETRIS_NO_CHAIN=1 puts every launch on one stream")
        if S == 1:
            algo_bytes = ALGO_BYTES[P] * N                            # per launch, per GPU: SURVEY §8(d) bytes x games
            achieved = algo_bytes / (launch_us * 1e-6) / 1e9
            if achieved > HBM_PEAK_GBS:
                roofline["frac_note"] = ("above 1: SURVEY 8(d)'s algorithmic bytes (192 B in + 192 B out per player-board) are a budget, not what the kernel "
                                         "moves, and the 20 MB state of 64k two-player games lives in the caches (the XCDs' L2s in the affine form, the 256 MB Infinity Cache otherwise)")
            roofline.update({"achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": algo_bytes,
                             "frac_kernel": (algo_bytes / (launch_us_events * 1e-6) / 1e9 / HBM_PEAK_GBS) if launch_us_events else None})
            if gpu_paced_us:
                roofline.update({"launch_us_gpu_paced": gpu_paced_us, "frac_gpu_paced": algo_bytes / (gpu_paced_us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                 "gpu_paced_note": ("NOT the timed kernel: the write-through kernel on HIP streams (the pre-queued call goes that way). " if affine else "") +
                                                   "extra run AFTER the timed region: 512 launches queued behind a blocker kernel before the first starts "
                                                   "(HIP events); `value`, `launch_us` and `frac` above are the host-paced wall clock of the K timed launches "
                                                   "(on a host that launches faster than the GPU steps the two agree to a few per cent; a slower host shows in the wall clock only)"})
            # HBM traffic cannot be measured inside this process: it comes from separate rocprofv3 --pmc passes over this same
            # command (profiles/pmc_passes.sh), corrected as MI355X_MICROARCH.md prescribes; the file is named next to the number
            pmc = os.path.join(ROOT, "profiles", "r03", f"pmc_p{P}_s{S}.json")
            traffic, source = None, None
            if affine:
                # the affine kernels keep the state in the XCDs' L2s for the length of a call: what reaches the fabric per launch is no
                # property of one dispatch (dirty lines leave at the call's last packet), and a --pmc run, whose dispatches are serialised with
                # the tool's own packets between them, measures another regime (profiles/r03/pmc_p1_affine_summary.json: WRITE_SIZE median
                # 36 KiB per dispatch).  The write-through kernel's measured figure is reported beside the null.
                source = ("none: state is L2-resident across the launches of a call (XCD-affine kernel); per-dispatch PMC counters do not describe it — "
                          "profiles/r03/pmc_p1_affine_summary.json; `traffic_write_through_kernel` is the measured figure of the non-affine kernel")
                if os.path.exists(pmc):
                    try:
                        roofline["traffic_write_through_kernel"] = json.load(open(pmc)).get("hbm_bytes_per_launch")
                    except Exception:
                        pass
            elif os.path.exists(pmc):
                try:
                    traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
                    source = (os.path.relpath(pmc, ROOT) + " (separate rocprofv3 --pmc passes of this command with the dispatches serialised, "
                              "TETRIS_CHAIN_DEPTH=1: raw TCC_EA0 request counters -> bytes, reads x the factor calibrated on zero-step launches; "
                              "fabric side of the L2, Infinity-Cache hits included)")
                except Exception:
                    traffic = None
            roofline.update({"traffic": traffic, "traffic_source": source})
        else:
            # a fused launch keeps the state in registers for S steps: SURVEY's per-step byte budget does not apply, and a
            # fraction of the HBM roofline would say nothing about it
            roofline.update({"achieved": None, "frac": None, "frac_kernel": None, "traffic": None,
                             "note": f"{S} env-steps fused per launch: state is loaded and stored once per launch, no per-step HBM roofline applies"})
        line = {
            "metric": "env-steps/sec at 64k parallel 20x10 boards",
            "value": env_steps / wall,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": wall * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": f"{N} parallel {args.height}x10 {'single' if P == 1 else 'two'}-player boards per GPU, "
                            f"random (rotation,translation) policy, auto-reset, {S} env-step(s) per launch",
                "games_per_gpu": N, "players": P, "steps_per_launch": S, "parallelism": f"replicas x{world} (no collective)",
            },
            "roofline": roofline,
            "env_steps_counted_on_device": env_steps,
            "episodes": int(counters[1]), "lines_cleared": int(counters[2]), "garbage_sent": int(counters[3]),
            "device_preconditioning": {"ms": args.precondition_ms, "launches_on_a_scratch_batch": precondition_launches,
                                       "note": "untimed, before the warm-up steps, on a second batch that is then discarded: brings the device clocks out of idle"},
            "library": os.path.relpath(lib_path, ROOT) if lib_path.startswith(ROOT) else lib_path,
            "device": capi.device_name(local_rank, os.environ.get("BENCH_LIB_PATH")),
        }
        if args.cpu_seconds > 0 and world == 1:
            line["cpu_baseline"] = cpu_baseline(P, args.height, args.cpu_seconds)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    batch.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
