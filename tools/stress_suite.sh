#!/usr/bin/env bash
# Reproduction attempts for round 5's dead GPUTEST (VERDICT r5 "next" 1 (iii)); everything is logged under gpurun_out/stress/.
#   bash tools/stress_suite.sh driver N          the driver's exact command, N times back to back (fresh process each)
#   bash tools/stress_suite.sh tail REPS         the step tests at the configs[3] / [4] shapes + the training files, REPS times in ONE process
#                                                 per file, kernels serialised (AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
O=gpurun_out/stress; mkdir -p $O
{ echo "== $(date -u +%FT%TZ) $(rocminfo 2>/dev/null | grep -m1 'Marketing Name')"; nproc; free -g | head -2; ulimit -a | grep -E "core|virtual|max memory|open files|processes"; } >> $O/box.txt 2>&1
sub=${1:-driver}; n=${2:-1}
case $sub in
driver)
  for i in $(seq 1 $n); do
    t0=$(date +%s)
    python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=25 > $O/driver_$i.log 2> $O/driver_$i.err
    rc=$?
    echo "driver run $i: rc=$rc $(( $(date +%s) - t0 ))s | $(grep -aE 'passed|failed' $O/driver_$i.log | tail -n 1)" | tee -a $O/summary.txt
    if [ $rc -ne 0 ]; then grep -a "\[gpu-test\]" $O/driver_$i.err | tail -n 2 | tee -a $O/summary.txt; grep -a -i -m5 "fault\|abort\|error" $O/driver_$i.err | tee -a $O/summary.txt; fi
  done ;;
tail)
  export AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1
  for f in tests/test_cfg_step_parity.py tests/test_train_gpu.py tests/test_train_update.py tests/test_train_encoder.py tests/test_trainer_gpu.py tests/test_train_data.py; do
    b=$(basename $f .py); t0=$(date +%s)
    args=""; for i in $(seq 1 $n); do args="$args $f"; done
    timeout 3000 python3 -m pytest $args -x -q -m gpu -p no:cacheprovider --keep-duplicates > $O/tail_$b.log 2> $O/tail_$b.err
    echo "tail x$n $b: rc=$? $(( $(date +%s) - t0 ))s | $(grep -aE 'passed|failed' $O/tail_$b.log | tail -n 1)" | tee -a $O/summary.txt
  done ;;
esac
