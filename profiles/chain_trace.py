"""Diagnostic: the timeline of chained launches (k_chain, 64k single-player boards, one env-step per launch).

Loads a -DTE_PHASE_TRACE build; every wave of the last 8 launches stamps the 100 MHz real-time clock at
entry / policy drawn / epoch seen / all state words arrived (forced wait) / step done, stores issued / stores acknowledged /
epoch published.  Prints per-phase medians and, per wave, the distance between consecutive launches' same phases (the period)
and the hand-off gap: publication by launch E-1 -> "epoch seen" by launch E.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DTE_PHASE_TRACE=1 -I include \
          -o profiles/_ab/lib_trace1.so drl-tetris_amd/csrc/tetris_hip.hip drl-tetris_amd/csrc/tetris_hip_multi.hip
    python profiles/chain_trace.py profiles/_ab/lib_trace1.so
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

pkg = ge.package()
lib_path = os.path.abspath(sys.argv[1])
N = 65536
b = pkg.TetrisBatch(N, 1, 20, 10, seeds=np.arange(N), lib_path=lib_path)
lib = C.CDLL(lib_path)
lib.tetris_debug_chain_trace.argtypes = [C.c_void_p]
lib.tetris_debug_trace.argtypes = [C.c_void_p, C.c_int]
buf2 = np.zeros(2048 * 16, np.uint64)
PH = {4: "philox+prefetch issued", 10: "kt: shapes+band+no-kick tests", 11: "kt: kick path", 12: "kt: slide", 13: "kt: hard drop", 5: "key interpreter done",
      6: "settle (clear/spawn) done", 7: "tick done", 8: "auto-reset done"}
PH_ORDER = [4, 10, 11, 12, 13, 5, 6, 7, 8]
buf = np.zeros(8 * 1024 * 8, np.uint64)
STREAMS = 3          # chain streams the library rotates over (CHAIN_STREAMS): launch E follows launch E - STREAMS on its stream
NAMES = ["entry", "policy drawn", "epoch seen", "state arrived", "stores issued", "stores acked", "published"]
b.rollout_random(200, 1)
assert b.rollout_is_chained(1)
for rep in range(3):
    _, ms = b.rollout_random(512, 1, first_step=1000 + 600 * rep)
    b.sync()
    assert lib.tetris_debug_chain_trace(buf.ctypes.data) == 0
    raw = buf.reshape(8, 1024, 8)
    place = raw[:, :, 7]
    t = raw.astype(np.int64)[:, :, :7] / 100.0          # us
    order = np.argsort(t[:, 0, 0])                                           # epochs by time
    t = t[order]
    place = place[order]
    xcc, hw = (place >> np.uint64(32)).astype(np.int64) & 15, place.astype(np.int64) & 0xFFFFFFFF
    simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    simd_key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    print(f"--- sample {rep}: {ms * 1e3 / 512:.2f} us per launch (stamped build)")
    e = slice(2, 7)                                                          # launches in the middle of the last 8
    for k in range(6):
        d = (t[e, :, k + 1] - t[e, :, k]).ravel()
        print(f"  {NAMES[k]:14s} -> {NAMES[k + 1]:14s} median {np.median(d):5.2f} us  p10 {np.percentile(d, 10):5.2f}  p90 {np.percentile(d, 90):5.2f}")
    life = (t[e, :, 6] - t[e, :, 0]).ravel()
    print(f"  wave lifetime median {np.median(life):.2f}  p90 {np.percentile(life, 90):.2f}")
    per = (t[3:7, :, 6] - t[2:6, :, 6]).ravel()
    print(f"  period per wave (published E -> published E+1): median {np.median(per):.2f}  p10 {np.percentile(per, 10):.2f}  p90 {np.percentile(per, 90):.2f}")
    gap = (t[3:7, :, 2] - t[2:6, :, 6]).ravel()
    print(f"  hand-off: published by E-1 -> epoch seen by E: median {gap.mean():.2f} mean, {np.median(gap):.2f} median, p90 {np.percentile(gap, 90):.2f}")
    early = (t[2:6, :, 6] - t[3:7, :, 1]).ravel()
    print(f"  successor already waiting when E-1 publishes (published E-1 minus 'policy drawn' of E): median {np.median(early):.2f}  p10 {np.percentile(early, 10):.2f}  frac>0 {np.mean(early > 0):.2f}")
    for k in range(1, 8 - STREAMS):
        first_in, last_pub = t[k + STREAMS, :, 0].min(), t[k, :, 6].max()
        med_pub = np.median(t[k, :, 6])
        w = int(np.argmax(t[k, :, 6]))
        ph = " ".join(f"{t[k, w, j + 1] - t[k, w, j]:.2f}" for j in range(6))
        print(f"  launch {k}: last publish {last_pub - t[k, :, 0].min():.2f} us after its first entry (median wave {med_pub - t[k, :, 0].min():.2f}); "
              f"same-stream successor enters {first_in - last_pub:.2f} us after that; slowest wave {w}: entry+{t[k, w, 0] - t[k, :, 0].min():.2f} phases {ph}")
    # placement: how many waves of ONE launch share a SIMD, and does sharing make a wave slow?
    for k in (3, 4):
        keys, inv, counts = np.unique(simd_key[k], return_inverse=True, return_counts=True)
        share = counts[inv]
        comp = t[k, :, 4] - t[k, :, 3]
        print(f"  launch {k}: {len(keys)} SIMDs used by 1024 waves; waves per SIMD histogram {np.bincount(counts).tolist()}; "
              + "; ".join(f"compute median with {c} on the SIMD: {np.median(comp[share == c]):.2f} us (n={int((share == c).sum())})" for c in sorted(set(share.tolist()))))
        print(f"     distinct XCDs {len(set(xcc[k].tolist()))}, CUs {len(set((simd_key[k] // 4).tolist()))}; same wave on the same SIMD as in the launch before: {np.mean(simd_key[k] == simd_key[k - 1]):.2f}, same XCD: {np.mean(xcc[k] == xcc[k - 1]):.2f}, same XCD as {STREAMS} launches before: {np.mean(xcc[k] == xcc[k - STREAMS]):.2f}")
    # inside the step of the LAST launch (shader-clock stamps of game_run, ~2.47 cycles per ns)
    assert lib.tetris_debug_trace(buf2.ctypes.data, buf2.size) == 0
    c = buf2.reshape(2048, 16)[:1024].astype(np.int64)
    print("  inside the step (last launch, shader clock / 2470 per us): " +
          "; ".join(f"{PH[a]} -> {PH[b_]}: {np.median((c[:, b_] - c[:, a]) / 2470.0):.2f}" for a, b_ in zip(PH_ORDER[:-1], PH_ORDER[1:])))
    # the slowest waves of the last launch: which part of the step took long?
    last = int(np.argmax(t[:, 0, 0]))
    comp = t[last, :, 4] - t[last, :, 3]
    worst = np.argsort(comp)[::-1][:8]
    keys = [k for k in PH_ORDER if (c[:, k] != 0).all()]
    print("  slowest waves of the last launch (state arrived -> stores issued, then the in-step phases in the order above; last column: tick done -> stores issued incl. reset):")
    for w in list(worst) + [-1]:
        if w < 0:
            ph = " ".join(f"{np.median((c[:, b_] - c[:, a]) / 2470.0):4.2f}" for a, b_ in zip(keys[:-1], keys[1:]))
            print(f"   median wave      {np.median(comp):4.2f} | {ph}")
        else:
            ph = " ".join(f"{(c[w, b_] - c[w, a]) / 2470.0:4.2f}" for a, b_ in zip(keys[:-1], keys[1:]))
            print(f"   wave {w:4d} (xcd {xcc[last, w]} cu {cu[last, w]:2d} simd {simd[last, w]}) {comp[w]:4.2f} | {ph}")
    span = t[e, :, 0].max(axis=1) - t[e, :, 0].min(axis=1)
    print(f"  dispatch ramp (first to last wave entry of one launch): {np.median(span):.2f} us;  launch-to-launch first entry: {np.median(np.diff(t[:, :, 0].min(axis=1))):.2f} us")
b.close()
