#!/bin/bash
# Copies the judged summaries of profiles/r03_final_a.sh / r03_final_b.sh runs from gpurun_out/final3a, final3b (scratch) into
# profiles/r03/final and refreshes profiles/r03/pmc_p{1,2}_s1.json (read by bench.py for roofline.traffic).
set -e
cd "$(dirname "$0")/.."
A=gpurun_out/final3a; B=gpurun_out/final3b; DST=profiles/r03/final
mkdir -p $DST
newest() { ls -t $(find "$1" -name "$2") | head -1; }
if [ -d $A ]; then
  cp $A/bench_p1_s1.json $A/bench_p1_s1_driver_flags.json $A/bench_p2_s1.json $A/bench_p1_s1_unchained.json $A/bench_p2_s1_unchained.json $A/bench_p1_s32.json $DST/
  cp $A/split_stages.json $A/dropin_api.json $A/dropin_breakdown.json $A/chain_period_from_trace_p1.json $A/chain_period_from_trace_p2.json $DST/
  cp $A/order_check_p1.txt $A/order_check_p2.txt $DST/
  cp $A/bench_p1_s1_write_through.json $A/bench_p1_s1_driver_flags_write_through.json $A/bench_p2_s1_write_through.json $A/chain_soak.txt $DST/ 2>/dev/null || true
  cp $A/bench_p1_s1_streams.json $A/bench_p1_s1_driver_flags_streams.json $A/bench_p2_s1_streams.json $A/driver_flags_ab.txt $A/direct_vs_streams.txt $DST/ 2>/dev/null || true
  cp $A/pytest_gpu.log $DST/pytest_gpu.txt; cp $A/smoke.log $DST/smoke.txt
  cp $A/prof_p1.log $DST/bench_under_profiler_p1_s1.txt; cp $A/prof_p2.log $DST/bench_under_profiler_p2_s1.txt
  cp "$(newest $A/prof_p1 '*kernel_stats.csv')" $DST/kernel_stats_p1_s1.csv
  cp "$(newest $A/prof_p2 '*kernel_stats.csv')" $DST/kernel_stats_p2_s1.csv
  cp "$(newest $A/prof_p1_unchained '*kernel_stats.csv')" $DST/kernel_stats_p1_s1_unchained.csv
  [ -d $A/prof_p1_affine ] && cp "$(newest $A/prof_p1_affine '*kernel_stats.csv')" $DST/kernel_stats_p1_s1_affine.csv
  cp "$(newest $A/prof_p2_unchained '*kernel_stats.csv')" $DST/kernel_stats_p2_s1_unchained.csv
  cp "$(newest $A/prof_split '*kernel_stats.csv')" $DST/kernel_stats_split.csv
  for c in enum_planar observe step_auto_1p step_auto_2p step_obs_1p step_obs_2p; do
    cp $A/kernel_$c.json $DST/; cp "$(newest $A/prof_$c '*kernel_stats.csv')" $DST/kernel_stats_$c.csv
  done
  python - <<'PY'
import csv, glob, json, statistics
f = glob.glob("gpurun_out/final3a/prof_split/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
out = {"source": "rocprofv3 --kernel-trace of profiles/split_stages.py 256 (profiles/r03_final_a.sh): both sides of 65 536 two-player games as two batches on one GPU; "
                 "the dispatches of a stage alternate side 0 / side 1; the first quarter of each kernel's dispatches (warm-up) is left out", "mean_duration_us": {}}
for k, name in (("k_split<0", "stage A"), ("k_split<1", "stage B"), ("k_split<2", "stage C"), ("k_split<3", "stages C + A of the next step")):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if k in r["Kernel_Name"]]
    d = d[len(d) // 4:]
    out["mean_duration_us"][k + ", false> (" + name + ")"] = {"side0": round(statistics.fmean(d[0::2]), 3), "side1": round(statistics.fmean(d[1::2]), 3), "dispatches": len(d)}
m = out["mean_duration_us"]
b, ca = m["k_split<1, false> (stage B)"], m["k_split<3, false> (stages C + A of the next step)"]
out["kernel_us_per_step_two_kernel_form"] = {"side0": round(b["side0"] + ca["side0"], 3), "side1": round(b["side1"] + ca["side1"], 3)}
json.dump(out, open("profiles/r03/final/split_stages_per_side.json", "w"), indent=1)
print(out["kernel_us_per_step_two_kernel_form"])
PY
fi
if [ -d $B ]; then
  for t in p1_serial p2_serial calib_p1 calib_p2 p1_three_streams p1_unchained p2_unchained; do cp $B/pmc_$t/summary.json $DST/pmc_${t}_summary.json; done
  python profiles/make_traffic_json.py $DST/pmc_p1_serial_summary.json 1 1 profiles/r03/pmc_p1_s1.json "k_chain<1>" $DST/pmc_calib_p1_summary.json $DST/pmc_p1_three_streams_summary.json
  python profiles/make_traffic_json.py $DST/pmc_p2_serial_summary.json 2 1 profiles/r03/pmc_p2_s1.json "k_duo<6, true" $DST/pmc_calib_p2_summary.json
  python - <<'PY'
import json
rows = []
for name, f, sub in (("k_chain<1> (one player, chained kernel, dispatches on one stream)", "pmc_p1_serial_summary.json", "k_chain<1>"),
                     ("k_game<1, 6, false> (one player, un-chained)", "pmc_p1_unchained_summary.json", "k_game<1, 6"),
                     ("k_duo<6, true> (two players, chained kernel, one stream)", "pmc_p2_serial_summary.json", "k_duo<6, true"),
                     ("k_duo<6, false> (two players, un-chained)", "pmc_p2_unchained_summary.json", "k_duo<6, false")):
    d = json.load(open("profiles/r03/final/" + f))
    k = [v for n, v in d.items() if sub in n][0]
    w = k["SQ_WAVES__stats"]["median"]
    g = lambda c: k[c + "__stats"]["median"] / w
    rows.append(f"{name:72s} {g('SQ_INSTS_VALU'):7.0f} {g('SQ_INSTS_SALU'):7.0f} {g('SQ_INSTS_VMEM_RD'):7.1f} {g('SQ_INSTS_VMEM_WR'):7.1f} {g('SQ_INSTS_LDS'):5.1f} "
                f"{g('SQ_WAVE_CYCLES'):9.0f} {g('SQ_ACTIVE_INST_ANY'):9.0f} {g('SQ_WAIT_ANY'):9.0f} {g('SQ_WAIT_INST_ANY'):9.0f}   {int(w)}")
open("profiles/r03/instruction_counts.txt", "w").write(
    "# executed instructions and SQ cycle counters per WAVE and launch (one env-step), medians over the dispatches of a run, rocprofv3 --pmc (separate passes),\n"
    "# 65 536 games; round 2's figures for k_game<1, 6>: 715 VALU + 233 SALU after that round's work (813 + 212 before): profiles/r02/instruction_counts.txt.\n"
    "# Round 3 changed no one-player kernel's instruction count on purpose; the key-interpreter rework (play_rt: columns shifted once, the second band window\n"
    "# derived from the first) left counts and times where they were (profiles/r03/play_rt_rows_ab.txt).  k_duo went from 199 to 115-130 registers per lane.\n"
    "# cycles are SQ quad-cycles (x 4 = shader cycles); a rocprofv3 --pmc run serialises dispatches, so a chained kernel's waves never wait for a predecessor here.\n"
    f"{'kernel':72s} {'VALU':>7s} {'SALU':>7s} {'VMEM_RD':>7s} {'VMEM_WR':>7s} {'LDS':>5s} {'WAVE_CYC':>9s} {'ACTIVE':>9s} {'WAIT_ANY':>9s} {'WAIT_INST':>9s}   waves\n" + "\n".join(rows) + "\n")
print(open("profiles/r03/instruction_counts.txt").read())
PY
fi
ls $DST | head -60
