#!/bin/bash
# quick feedback while tuning the step: parity at full size, chained / un-chained period, executed instructions per wave
set -e -o pipefail
O=gpurun_out/inst_now
mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python tests/tools/chain_parity.py | tail -1
python profiles/order_check.py 1 | tail -6
(cd /tmp; export TMPDIR=/tmp; timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS --kernel-trace --output-format csv -d $R/$O/pmc -- python3 $R/profiles/abl_run.py full > $R/$O/pmc.log 2>&1)
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/pmc/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_game" in row["Kernel_Name"]:
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
m = {k: a[0] / max(1, a[1]) for k, a in acc.items()}
w = m.get("SQ_WAVES", 1) or 1
print("per wave: VALU %.0f SALU %.0f VMEM_RD %.0f VMEM_WR %.0f LDS %.0f" % tuple(m.get(k, 0) / w for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS")))
PY
rm -rf $O/pmc
