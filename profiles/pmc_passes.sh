#!/bin/bash
# rocprofv3 PMC passes (run on the GPU box through gpurun).  One counter group per pass, counters only with
# --kernel-trace (MI355X_MICROARCH.md §rocprofv3 PMC slots; gpurun refuses --pmc combined with sys/hip/hsa traces).
#   usage: profiles/pmc_passes.sh <outdir> <mem|all> <script.py> [script args...]
# The TCC groups collect the RAW request counters (4 TCC slots per pass) that FETCH_SIZE / WRITE_SIZE are derived from, and the
# derived counters themselves as a cross-check.
OUT=$1; MODE=$2; SCRIPT=$3; shift; shift; shift
ROOTDIR=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOTDIR/$OUT
cd /tmp; export TMPDIR=/tmp
CNT_MEM=("TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum" "WRITE_SIZE" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum")
CNT_SQ=("SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
        "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE")
if [ "$MODE" = "all" ]; then CNT=("${CNT_SQ[@]}" "${CNT_MEM[@]}"); elif [ "$MODE" = "sq" ]; then CNT=("${CNT_SQ[@]}"); else CNT=("${CNT_MEM[@]}"); fi
i=0
for grp in "${CNT[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $ROOTDIR/$OUT/pass$i -- python3 $ROOTDIR/$SCRIPT "$@" > $ROOTDIR/$OUT/pass$i.log 2>&1 || echo "pass $i ($grp) failed: $(tail -2 $ROOTDIR/$OUT/pass$i.log)"
done
cd $ROOTDIR
python3 profiles/pmc_summary.py $OUT
