#!/bin/bash
# executed instructions per wave of the step kernel, full and with one phase compiled out (-DTE_ABLATE), from SQ counters
set -e -o pipefail
O=gpurun_out/${OUT:-r02i}
mkdir -p $O
R=$GRAFT_REPO_ROOT
for v in full 1 2 4 8 16 31; do
  lib=full; [ $v != full ] && lib=$R/drl-tetris_amd/lib/abl_$v.so
  python profiles/abl_run.py $lib > $O/time_$v.txt
  (cd /tmp; export TMPDIR=/tmp; timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$O/pmc_$v -- python3 $R/profiles/abl_run.py $lib > $R/$O/pmc_$v.log 2>&1)
done
python - <<PY
import csv, glob, collections
names = {"full": "full step", "1": "- Philox policy draw", "2": "- rotations / slide (key interpreter without them)", "4": "- delayCheck (timers, combo)", "8": "- clear + spawn", "16": "- auto-reset", "31": "- all five (load, hard drop + stamp, store remain)"}
print("%-56s %8s %8s %8s %8s %8s %10s" % ("build", "VALU", "SALU", "VMEM_RD", "VMEM_WR", "LDS", "us/launch"))
for v in ["full", "1", "2", "4", "8", "16", "31"]:
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob("$O/pmc_%s/**/*counter_collection.csv" % v, recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_game" in row["Kernel_Name"]:
                a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
    m = {k: a[0] / max(1, a[1]) for k, a in acc.items()}
    w = m.get("SQ_WAVES", 1) or 1
    t = open("$O/time_%s.txt" % v).read().split()[0]
    print("%-56s %8.0f %8.0f %8.0f %8.0f %8.0f %10s" % (names[v], m.get("SQ_INSTS_VALU", 0) / w, m.get("SQ_INSTS_SALU", 0) / w, m.get("SQ_INSTS_VMEM_RD", 0) / w, m.get("SQ_INSTS_VMEM_WR", 0) / w, m.get("SQ_INSTS_LDS", 0) / w, t))
PY
