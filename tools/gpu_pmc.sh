#!/usr/bin/env bash
# PMC study of one kernel group.  usage: bash tools/gpu_pmc.sh <which> <policy> <tag>
set -u
W=${1:-gru}; POL=${2:-mixed}; TAG=${3:-pmc}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o k -- python $REPO/tools/run_kernel.py $W $POL > /dev/null 2> $OUT/trace.err
python $REPO/tools/kstats.py $(find $OUT/trace -name "*kernel_stats.csv" | head -1) 12
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -o k -- python $REPO/tools/run_kernel.py $W $POL > /dev/null 2> $OUT/pmc$i.err
  f=$(find $OUT/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $REPO/tools/kstats.py $f | grep -A12 -E "k_gemm|k_attn|k_corr|k_conv|k_pv|k_flash" | head -40; else tail -3 $OUT/pmc$i.err; fi
done
rm -rf $OUT/trace $OUT/pmc*/ 2>/dev/null
