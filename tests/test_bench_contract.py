"""bench.py's output contract, rehearsed without GPUs: BENCH_LIB_PATH points the binding at the CPU harness build (test-only)
and BENCH_DIST_BACKEND=gloo replaces RCCL; everything else is the code path the driver runs — one process per rank under
torch.distributed.run, barrier + max-over-ranks timing, ONE JSON line from rank 0 with whole-job throughput."""
import json
import os
import subprocess
import sys

import __graft_entry__ as ge

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"]


def _run(cmd, extra_env):
    env = dict(os.environ, BENCH_LIB_PATH=ge.build_harness(), **extra_env)
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]            # rank 0 only, exactly one line
    return json.loads(lines[0])


def _check(line, world, games, steps, warmup, spl):
    for k in REQUIRED:
        assert k in line, k
    assert line["n_gpus"] == world and line["steps"] == steps and line["warmup"] == warmup
    assert line["scaling"] == "weak" and line["higher_is_better"] is True and line["vs_baseline"] is None
    assert line["unit"] == "env-steps/s" and line["data"] == "synthetic" and line["dtype"] == "u32"
    assert "workload" in line["config"] and "model" not in line["config"]
    # whole-job aggregate: all ranks' env-steps over the max-over-ranks time of exactly `steps` launches
    assert abs(line["value"] - world * games * spl / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
    assert line["env_steps_counted_on_device"] == world * games * spl * steps
    assert line["library"].endswith(".so") and line["device"]         # a BENCH_LIB_PATH run identifies itself
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    if spl == 1:
        # value and roofline.frac come from ONE clock: value x 389 B / 8 TB/s IS the fraction
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
        assert r["algorithmic_bytes_per_launch"] == 389 * games
        assert abs(r["frac"] - line["value"] / world * 389 / 8e12) <= 1e-9 * r["frac"]
    else:
        assert r["frac"] is None and r["achieved"] is None           # fused launches: no per-step HBM roofline


def test_single_process_line_with_cpu_baseline():
    line = _run([sys.executable, "bench.py", "--games", "1024", "--steps", "6", "--warmup", "2", "--cpu-seconds", "0.3"], {})
    _check(line, 1, 1024, 6, 2, 1)
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("reference", "port") and cb["unit"] == "env-steps/s" and cb["sample"]


def test_two_ranks_under_torch_distributed_run():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", "bench.py", "--gpus", "2", "--games", "1024", "--steps", "5", "--warmup", "1",
           "--steps-per-launch", "2", "--cpu-seconds", "0"]
    line = _run(cmd, {"BENCH_DIST_BACKEND": "gloo"})
    _check(line, 2, 1024, 5, 1, 2)
    assert line["cpu_baseline"] is None                # rank 0 at N=1 only


def test_plain_gpus_2_starts_its_own_two_ranks():
    """`python bench.py --gpus 2` with NO launcher: the script spawns its ranks itself and rank 0's single line says n_gpus 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_LIB_PATH=ge.build_harness(), BENCH_DIST_BACKEND="gloo")
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--games", "1024", "--steps", "5", "--warmup", "1", "--cpu-seconds", "0"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    _check(json.loads(lines[0]), 2, 1024, 5, 1, 1)


def test_plain_gpus_4_split_workload_runs_two_pairs():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(BENCH_LIB_PATH=ge.build_harness(), BENCH_DIST_BACKEND="gloo")
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--workload", "split", "--games", "512", "--steps", "4", "--warmup", "1"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["env_steps_counted_on_device"] == 2 * 512 * 4 and "pair" in line["config"]["parallelism"]


def test_more_ranks_than_devices_is_an_error_not_a_smaller_run():
    """Without the rehearsal library the parent counts the visible GPUs before it starts anything: asking for more is rc != 0
    and no JSON line (a run on fewer GPUs that calls itself --gpus N would poison a scaling curve)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "BENCH_LIB_PATH")}
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "64", "--steps", "5", "--warmup", "1", "--cpu-seconds", "0"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "GPU(s) visible" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_world_size_that_differs_from_gpus_is_an_error():
    env = dict(os.environ, BENCH_LIB_PATH=ge.build_harness(), BENCH_DIST_BACKEND="gloo", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--games", "256", "--steps", "2", "--warmup", "0", "--cpu-seconds", "0"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr
