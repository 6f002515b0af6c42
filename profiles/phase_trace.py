"""Diagnostic: where does a 64k-board launch spend its time?

Loads a -DTE_PHASE_TRACE build (stamps the shader clock at phase boundaries, first active lane of every wave, see
tetris_engine.h:te_stamp) and prints, over all 1024 waves of ONE launch in the middle of a back-to-back stream:
  * the dispatch ramp: when waves start / end relative to the first wave of the launch (100 MHz real-time clock);
  * per-phase durations in shader-clock cycles (median / p10 / p90 over waves).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DTE_PHASE_TRACE=1 -I include \
          -o profiles/_ab/lib_trace1.so drl-tetris_amd/csrc/tetris_hip.hip drl-tetris_amd/csrc/tetris_hip_multi.hip
    python profiles/phase_trace.py profiles/_ab/lib_trace1.so [P [S]]
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.package()
lib_path = os.path.abspath(sys.argv[1])
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # fused steps per launch: the in-loop stamps then show the LAST step
os.environ["TETRIS_NO_DUO"] = "1"
N = 65536
b = pkg.TetrisBatch(N, P, 20, 10, seeds=np.arange(N), lib_path=lib_path)
lib = C.CDLL(lib_path)
lib.tetris_debug_trace.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(2048 * 16, np.uint64)

NAMES = {0: "entry", 2: "loads issued", 3: "all loads arrived (forced wait)", 4: "philox+prefetch issued", 5: "key interpreter done",
         6: "settle (clear/spawn) done", 7: "tick done", 8: "auto-reset done", 9: "stores issued", 14: "exit",
         10: "kt: shapes+band+no-kick tests", 11: "kt: kick path", 12: "kt: slide", 13: "kt: hard drop"}
ORDER = [0, 2, 3, 4, 10, 11, 12, 13, 5, 6, 7, 8, 9, 14]

b.rollout_random(200, 1)                       # warm, tables extended
for rep in range(3):
    b.rollout_random(64, S, first_step=1000 + 100 * S * rep)      # the stamps of the LAST launch survive (each launch overwrites)
    b.sync()
    assert lib.tetris_debug_trace(buf.ctypes.data, buf.size) == 0
    t = buf.reshape(2048, 16)[: N // 64].astype(np.int64)
    rt0, rt1 = t[:, 1], t[:, 15]
    base = rt0.min()
    start_us = (rt0 - base) / 100.0
    end_us = (rt1 - base) / 100.0
    print(f"--- launch sample {rep}: {len(t)} waves")
    print(f"wave start (us after first wave): median {np.median(start_us):.2f}  p90 {np.percentile(start_us, 90):.2f}  max {start_us.max():.2f}")
    print(f"wave end   (us after first wave): median {np.median(end_us):.2f}  p10 {np.percentile(end_us, 10):.2f}  max {end_us.max():.2f}")
    print(f"wave lifetime us: median {np.median(end_us - start_us):.2f}  p10 {np.percentile(end_us - start_us, 10):.2f}  p90 {np.percentile(end_us - start_us, 90):.2f}")
    cyc_per_us = np.median((t[:, 14] - t[:, 0]) / np.maximum(end_us - start_us, 0.01))
    print(f"shader clock ~{cyc_per_us:.0f} cycles/us")
    keys = [k for k in ORDER if (t[:, k] != 0).all()]
    for a, c in zip(keys[:-1], keys[1:]):
        d = (t[:, c] - t[:, a]) / cyc_per_us
        print(f"  {NAMES[a]:36s} -> {NAMES[c]:36s} median {np.median(d):5.2f} us  p10 {np.percentile(d, 10):5.2f}  p90 {np.percentile(d, 90):5.2f}")
    # the launch ends with its slowest wave: what did the last finishers spend their time on?
    order = np.argsort(end_us)[::-1][:6]
    print("  slowest waves (wave id: start, end us; then per-phase us in the order above):")
    for w in order:
        ph = " ".join(f"{(t[w, c] - t[w, a]) / cyc_per_us:4.2f}" for a, c in zip(keys[:-1], keys[1:]))
        print(f"   wave {w:4d}: {start_us[w]:4.2f} {end_us[w]:4.2f} | {ph}")
    med = " ".join(f"{np.median((t[:, c] - t[:, a]) / cyc_per_us):4.2f}" for a, c in zip(keys[:-1], keys[1:]))
    print(f"   median wave          | {med}")
b.close()
