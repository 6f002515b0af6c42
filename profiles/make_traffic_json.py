#!/usr/bin/env python3
"""profiles/<round>/pmc_p{P}_s{S}.json (read by bench.py for roofline.traffic) from pmc_passes.sh summaries.

usage: make_traffic_json.py <summary.json of a SERIALISED run> <P> <S> <out.json> <kernel substring> <calib summary.json> [overlapped summary.json]

Bytes at the L2's memory side (fabric), per launch, from the RAW request counters with the expressions ROCm 7.2's counter_defs.yaml
gives for gfx950 (the medians over the run's dispatches):
    write = TCC_EA0_WRREQ_64B * 64 + (TCC_EA0_WRREQ - TCC_EA0_WRREQ_64B) * 32                     (= WRITE_SIZE * 1024)
    read  = TCC_BUBBLE * 128 + (TCC_EA0_RDREQ - TCC_BUBBLE - TCC_EA0_RDREQ_32B) * 64 + TCC_EA0_RDREQ_32B * 32   (= FETCH_SIZE * 1024)
The read figure is multiplied by a factor CALIBRATED in the same GPU call on zero-step launches of the same kernel, whose traffic is
known exactly (MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports half the bytes of a coalesced streaming read; other access
widths must be calibrated on a known byte count): factor = known read bytes / read figure of the calibration run.
The write figure must not be below what the kernel provably stores (stored words x 4 B x games): the script REFUSES such a
summary (exit 1).
Infinity-Cache hits are inside these fabric-side counters (the 10 MB state of 64k boards is resident there): HBM proper sees less."""
import json
import sys

summary, P, S, out, want, calib = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6]
overlapped = sys.argv[7] if len(sys.argv) > 7 else None
GAMES = 65536
STORED_WORDS = {1: 25 + 3, 2: 2 * 27 + 3}       # words every launch stores per game: hot board words + game words


def med(k, c):
    return k[c + "__stats"]["median"]


def pick(d, sub):
    names = [n for n in d if sub in n]
    if not names:
        sys.exit(f"no kernel matching {sub!r} in {list(d)}")
    return names[0], d[names[0]]


def fabric_bytes(k):
    wr, wr64 = med(k, "TCC_EA0_WRREQ_sum"), med(k, "TCC_EA0_WRREQ_64B_sum")
    rd, rd32, bub = med(k, "TCC_EA0_RDREQ_sum"), med(k, "TCC_EA0_RDREQ_32B_sum"), med(k, "TCC_BUBBLE_sum")
    return wr64 * 64 + (wr - wr64) * 32, bub * 128 + (rd - bub - rd32) * 64 + rd32 * 32


d = json.load(open(summary))
name, k = pick(d, want)
cal = json.load(open(calib))
# the zero-step calibration launches: the same kernel for one player; for two players the un-chained one-lane kernel k_game<2, 6>
# (a zero-step launch is not a k_duo launch) — the same coalesced dword-per-lane rows, the same words
cname, ck = pick(cal, want.split("<")[0] if any(want.split("<")[0] in n for n in cal) else f"k_game<{P}, 6")
floor = STORED_WORDS[P] * 4 * GAMES
write, read_raw = fabric_bytes(k)
cwrite, cread_raw = fabric_bytes(ck)
# zero-step launches: every stored word was loaded (the step's table / start-word reads do not happen), plus one epoch word per wave
known_read = (STORED_WORDS[P] - (1 if P == 1 else 0)) * 4 * GAMES       # (1-player kernels store W_MIN_REMAINING without loading it)
factor = known_read / cread_raw
res = {
    "kernel": name, "steps_per_launch": S, "games": GAMES, "players": P,
    "dispatches_serialised": True, "dispatches": k["_dispatches"],
    "write_bytes": int(write), "write_bytes_floor_stored_words": floor,
    "WRITE_SIZE_KiB_median": med(k, "WRITE_SIZE") if "WRITE_SIZE__stats" in k else None,
    "read_bytes_raw": int(read_raw), "read_factor_calibrated": factor,
    "FETCH_SIZE_KiB_median": med(k, "FETCH_SIZE") if "FETCH_SIZE__stats" in k else None,
    "calibration": {"kernel": cname, "write_bytes": int(cwrite), "read_bytes_raw": int(cread_raw), "known_read_bytes": known_read,
                    "known_write_bytes": floor, "source": calib},
    "hbm_bytes_per_launch": int(write + factor * read_raw),
    "counters_median": {c[:-7]: v["median"] for c, v in k.items() if c.endswith("__stats")},
    "counters_min_max": {c[:-7]: [v["min"], v["max"]] for c, v in k.items() if c.endswith("__stats")},
    "source": summary,
}
if overlapped:
    oname, ok = pick(json.load(open(overlapped)), "k_chain" if P == 1 else "k_duo")
    res["overlapped_run_for_comparison"] = {
        "kernel": oname, "source": overlapped, "note": "the same command on three chain streams (the default): rocprofv3 serialises the dispatches of a --pmc run, so the counters agree with the one-stream run",
        "counters": {c[:-7]: {s: v[s] for s in ("n", "min", "median", "max", "mean")} for c, v in ok.items() if c.endswith("__stats") and c.startswith("TCC")}}
if write < floor:
    json.dump(res, open(out + ".refused", "w"), indent=1)
    sys.exit(f"REFUSED: write figure {int(write)} B per launch is below the {floor} B the kernel provably stores ({STORED_WORDS[P]} words x 4 B x {GAMES} games)")
if abs(cwrite - floor) > 0.02 * floor:
    sys.exit(f"REFUSED: the calibration run wrote {int(cwrite)} B, expected {floor} B")
json.dump(res, open(out, "w"), indent=1)
print(out, res["hbm_bytes_per_launch"], "write", int(write), "read", int(factor * read_raw), "factor", round(factor, 3))
