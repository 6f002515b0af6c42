#!/bin/bash
# Where do the wave-cycles of the step kernel go?  SQ counters of k_game<1, M_ROLLOUT> (64k single-player boards, one env-step per
# launch, un-chained), one rocprofv3 --pmc pass per group.  SQ_WAIT_ANY + SQ_WAIT_INST_ANY + SQ_ACTIVE_INST_ANY ~ SQ_WAVE_CYCLES
# (quad-cycles, MI355X_MICROARCH.md).
set -o pipefail
O=gpurun_out/${OUT:-stall}
mkdir -p $O
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVES SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_IFETCH SQ_WAVE_CYCLES"; do
  i=$((i+1))
  (cd /tmp; export TMPDIR=/tmp; timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/$O/pmc_$i -- python3 $R/profiles/abl_run.py full > $R/$O/pmc_$i.log 2>&1) || echo "group $i failed"
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_game" in row["Kernel_Name"]:
            a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
m = {k: a[0] / max(1, a[1]) for k, a in acc.items()}
w = m.get("SQ_WAVES", 1) or 1
print("per wave of one launch (%d waves):" % w)
for k in sorted(m):
    if k != "SQ_WAVES": print("  %-28s %10.1f" % (k, m[k] / w))
PY
rm -rf $O/pmc_*/
