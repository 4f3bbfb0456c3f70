#!/usr/bin/env bash
# Round 5: north_star's per-kernel rocprof evidence for the CURRENT tree (VERDICT r4 row (d'')).  One --pmc pass per counter group
# (never combined with trace domains), each over tools/run_kernel.py <group> at the bench shape (448x1024, batch 4).
#   bash tools/pmc_r5.sh <tag> [groups...]      -> gpurun_out/<tag>/<group>_<pass>.txt (+ kernel-trace stats of the same command)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
TAG=${1:-pmc_r5}; shift || true
GROUPS_=${*:-"grustep convtok menc flash fnet cnet corr pv probs"}
O=$REPO/gpurun_out/$TAG; mkdir -p $O
SQ1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
cd /tmp
for g in $GROUPS_; do
  cmd="python $REPO/tools/run_kernel.py $g mixed"
  i=0
  for ctrs in "FETCH_SIZE" "WRITE_SIZE" "$SQ1" "$SQ2"; do
    i=$((i + 1))
    d=$O/p_${g}_$i
    REPS=4 timeout 600 rocprofv3 --pmc $ctrs --output-format csv -d $d -o k -- $cmd > /dev/null 2> $O/${g}_$i.err
    f=$(find $d -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python $REPO/tools/kstats.py "$f" > $O/${g}_pass$i.txt; rm -f $O/${g}_$i.err; else tail -3 $O/${g}_$i.err; fi
    rm -rf $d
  done
  d=$O/kt_$g
  REPS=4 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o k -- $cmd > $O/${g}_out.txt 2> /dev/null
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && python $REPO/tools/kstats.py "$f" 30 > $O/${g}_kstats.txt
  rm -rf $d
  echo "== $g done"; head -12 $O/${g}_kstats.txt
done
