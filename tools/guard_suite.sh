#!/usr/bin/env bash
# Guard-page run of the GPU suite, one pytest process per test file (a GPU fault aborts the process: the other files still run).
#   bash tools/guard_suite.sh <mode: end|start> <per-file timeout s> [test files ...]      -> gpurun_out/guard_<mode>/
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
MODE=${1:-end}; TMO=${2:-600}; shift 2 || true
FILES=${@:-$(ls tests/test_cfg_step_parity.py tests/test_train_*.py tests/test_trainer_gpu.py tests/test_hip_ops.py tests/test_hip_e2e.py tests/test_gemm_pk*.py tests/test_full_size_parity.py tests/test_fuzz_parity.py tests/test_augment.py tests/test_eval_harness.py tests/test_harness_golden.py tests/test_integration_doc.py tests/test_reference_wrappers.py tests/test_bench_contract.py)}
O=gpurun_out/guard_$MODE; mkdir -p $O
for f in $FILES; do
  b=$(basename $f .py)
  t0=$(date +%s)
  timeout $TMO python tools/guard_run.py --mode $MODE --log $O/$b.last_call.txt -- $f -m gpu -q --tb=short -p no:cacheprovider --maxfail=10 > $O/$b.log 2>&1
  rc=$?
  echo "$b rc=$rc $(( $(date +%s) - t0 ))s $(tail -n 1 $O/$b.log | cut -c1-150)" | tee -a $O/summary.txt
  if [ $rc -ne 0 ] && [ -f $O/$b.last_call.txt ]; then echo "   last call: $(cut -c1-600 $O/$b.last_call.txt)" | tee -a $O/summary.txt; grep -a -m3 -i "memory access fault\|gpu-test\] " $O/$b.log | tail -n 3 | tee -a $O/summary.txt; grep -a "\[gpu-test\]" $O/$b.log | tail -n 1 | tee -a $O/summary.txt; fi
done
