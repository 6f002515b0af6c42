"""PCIe-inclusive rate of the host-buffer boundary (NOT the headline number): tetris_step_rt with numpy inputs/outputs and
tetris_step_keys (perform_action shape), 64k games, one synchronous call per env-step."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.package()
out = {}
for P in (1, 2):
    n = 65536
    b = pkg.TetrisBatch(n, P, 20, 10, seeds=np.arange(n))
    rng = np.random.default_rng(0)
    rot, trans = rng.integers(0, 4, n).astype(np.uint8), rng.integers(0, 10, n).astype(np.uint8)
    pl = np.zeros(n, np.uint8)
    for _ in range(5):
        b.step_rt(rot, trans, pl)
    t0 = time.perf_counter(); reps = 200
    for s in range(reps):
        done = b.step_rt(rot, trans, pl)
        idx = np.nonzero(done)[0].astype(np.int32)
        if len(idx):
            b.reset(idx, seeds=s)
    dt = (time.perf_counter() - t0) / reps
    keys = np.zeros((n, P, 14), np.uint8); lens = np.ones((n, P), np.uint8)
    keys[:, 0, :4] = [8, 2, 3, 7]; lens[:, 0] = 4
    t0 = time.perf_counter()
    for s in range(50):
        done, _, _ = b.step_keys(keys, lens)
        idx = np.nonzero(done)[0].astype(np.int32)
        if len(idx):
            b.reset(idx, seeds=s)
    dk = (time.perf_counter() - t0) / 50
    out[f"{P}p"] = {"step_rt_host_us_per_call_incl_reset": dt * 1e6, "step_rt_env_steps_per_s": n / dt,
                    "step_keys_host_us_per_call_incl_reset": dk * 1e6, "step_keys_env_steps_per_s": n / dk}
    b.close()
print(json.dumps(out))
