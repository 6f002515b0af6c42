#!/usr/bin/env python3
"""profiles/<round>/pmc_p{P}_s{S}.json (read by bench.py for roofline.traffic) from a pmc_passes.sh summary.
usage: make_traffic_json.py <summary.json> <P> <S> <out.json> [kernel-name substring]
HBM-side bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB: WRITE_SIZE is exact and FETCH_SIZE reads 1/2 on
gfx950 for this kernel's access pattern — calibrated with profiles/calib.py (zero-step launches moving exactly
29 words x 4 B x 65536 games = 7424 KiB written, 28 read: WRITE_SIZE 7424 KiB exact, FETCH_SIZE 3610 KiB = 1/2 of the 7168 KiB + kernel arguments), as
MI355X_MICROARCH.md §HBM prescribes.  The x2 is calibrated for the coalesced state stream; for the scattered
one-byte RNG-table reads it is an upper bound.  Infinity-Cache hits are included in these fabric-side counters."""
import json, sys
summary, P, S, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
d = json.load(open(summary))
want = sys.argv[5] if len(sys.argv) > 5 else None
names = [name for name in d if (want in name if want else (f"k_game<{P}, 6" in name or "k_chain" in name or "k_duo<6" in name))]
k = d[names[0]]
res = {
    "kernel": names[0], "steps_per_launch": S, "games": 65536,
    "FETCH_SIZE_KiB_raw": k["FETCH_SIZE"], "WRITE_SIZE_KiB": k["WRITE_SIZE"],
    "fetch_correction": 2.0,
    "hbm_bytes_per_launch": int((2.0 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024),
    "TCC_HIT_sum": k.get("TCC_HIT_sum"), "TCC_MISS_sum": k.get("TCC_MISS_sum"),
    "TCC_EA0_RDREQ_sum": k.get("TCC_EA0_RDREQ_sum"), "TCC_EA0_WRREQ_sum": k.get("TCC_EA0_WRREQ_sum"),
    "source": summary,
}
for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_WAVES", "SQ_WAVE_CYCLES",
          "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "GRBM_GUI_ACTIVE"):
    if c in k:
        res[c] = k[c]
json.dump(res, open(out, "w"), indent=1)
print(out, res["hbm_bytes_per_launch"])
