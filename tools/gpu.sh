#!/usr/bin/env bash
# The ONE GPU-box wrapper (run through gpurun from the build container); everything is logged under gpurun_out/.
#   bash tools/gpu.sh check [pytest args]            parity tests + smoke + short bench
#   bash tools/gpu.sh ab [bench args]                same-box A/B of two builds: craft_amd/libcraft_hip_prev.so vs libcraft_hip.so
#   bash tools/gpu.sh envab KNOB=1 [bench args]      same-box A/B of an environment knob
#   bash tools/gpu.sh kstats <tag> <cmd...>          rocprofv3 --kernel-trace --stats of <cmd> -> gpurun_out/<tag>/kernel_stats.txt
#   bash tools/gpu.sh pmc <tag> "<counters>" <cmd...>   one rocprofv3 --pmc pass of <cmd> -> gpurun_out/<tag>/pmc_<first counter>.txt
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $REPO/gpurun_out
val() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'])"; }
sub=${1:-check}; shift || true
case $sub in
check)
  cd $REPO
  rocminfo 2>/dev/null | grep -m3 -E "Marketing Name|gfx" > gpurun_out/gpu.txt; nproc >> gpurun_out/gpu.txt
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider "$@" > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit: $?" | tee -a gpurun_out/pytest_gpu.log; tail -n 40 gpurun_out/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?" | tee -a gpurun_out/smoke.log
  tail -n 3 gpurun_out/smoke.log
  timeout 900 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit: $?"; tail -n 5 gpurun_out/bench.err; cat gpurun_out/bench.log ;;
ab)
  cd $REPO
  for rep in 1 2 3; do for lib in libcraft_hip_prev.so libcraft_hip.so; do
    echo "$lib $(CRAFT_HIP_LIB=$REPO/craft_amd/$lib python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-train-leg "$@" 2>/dev/null | val)"
  done; done ;;
envab)
  cd $REPO; KNOB=$1; shift
  for r in 1 2 3; do
    echo "round $r: base $(python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg "$@" 2>/dev/null | val)   $KNOB $(env $KNOB python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg "$@" 2>/dev/null | val)"
  done ;;
kstats)
  TAG=$1; shift; O=$REPO/gpurun_out/$TAG; mkdir -p $O; cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- "$@" > $O/cmd_out.txt 2> $O/cmd_err.txt
  f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
  if [ -n "$f" ]; then cp "$f" $O/kernel_stats_full.csv; python $REPO/tools/kstats.py "$f" 70 > $O/kernel_stats.txt; head -50 $O/kernel_stats.txt; else tail -5 $O/cmd_err.txt; fi
  rm -rf $O/kt; tail -1 $O/cmd_out.txt | cut -c1-400 ;;
pmc)
  TAG=$1; CTRS=$2; shift 2; O=$REPO/gpurun_out/$TAG; mkdir -p $O; cd /tmp
  first=${CTRS%% *}
  timeout 900 rocprofv3 --pmc $CTRS --output-format csv -d $O/p_$first -o k -- "$@" > /dev/null 2> $O/pmc_$first.err
  f=$(find $O/p_$first -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $REPO/tools/kstats.py "$f" > $O/pmc_$first.txt; else tail -5 $O/pmc_$first.err; fi
  rm -rf $O/p_$first ;;
*) echo "unknown subcommand $sub"; exit 2 ;;
esac
