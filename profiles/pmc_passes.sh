#!/bin/bash
# rocprofv3 PMC passes for the step kernel (run on the GPU box through gpurun).  One counter group per
# pass, counters only with --kernel-trace (MI355X_MICROARCH.md §rocprofv3 PMC slots; gpurun refuses
# --pmc combined with sys/hip/hsa traces).   usage: profiles/pmc_passes.sh <outdir> [bench args...]
set -e
OUT=$1; shift
ROOTDIR=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOTDIR/$OUT
cd /tmp; export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $ROOTDIR/$OUT/pass$i -- python3 $ROOTDIR/bench.py --cpu-seconds 0 "$@" > $ROOTDIR/$OUT/pass$i.log 2>&1 || echo "pass $i failed"
done
cd $ROOTDIR
python3 profiles/pmc_summary.py $OUT
