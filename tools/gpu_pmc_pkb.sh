# PMC counters of craft_gemm_pk's kernels at the attention shapes of configs[3] (tools/bench_gemm_pkb.py)
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/pmc_pkb
mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python $REPO/tools/bench_gemm_pkb.py ${1:-f16x3} > $O/bench.txt 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && grep -E "k_gemm_pkb|Name" "$f" | cut -c1-200 > $O/kernel_stats.txt
rm -rf $O/kt
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -o k -- python $REPO/tools/bench_gemm_pkb.py ${1:-f16x3} > /dev/null 2> $O/pmc_sq.err
f=$(find $O/pmc_sq -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python $REPO/tools/kstats.py "$f" | grep -A10 "k_gemm_pkb" > $O/pmc_sq.txt
rm -rf $O/pmc_sq
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM --output-format csv -d $O/pmc_b -o k -- python $REPO/tools/bench_gemm_pkb.py ${1:-f16x3} > /dev/null 2> $O/pmc_b.err
f=$(find $O/pmc_b -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python $REPO/tools/kstats.py "$f" | grep -A8 "k_gemm_pkb" > $O/pmc_b.txt
rm -rf $O/pmc_b
cat $O/kernel_stats.txt $O/pmc_sq.txt $O/pmc_b.txt
tail -3 $O/pmc_b.err
