#!/usr/bin/env bash
# rocprofv3 kernel-trace stats of the bench command + PMC passes (separate runs) for the dominant kernel.
# usage: gpurun -- 'bash tools/gpu_profile.sh <tag> [bench args]'
set -u
TAG=${1:-r1}; shift || true
REPO=$(pwd); OUT=$REPO/gpurun_out/prof_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_under_rocprof.log 2> $OUT/rocprof.err
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -30 $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -o pv -- python $REPO/tools/run_kernel.py pv mixed > /dev/null 2> $OUT/pmc_$c.err
  f=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
  echo "== $c ($f)"; grep -E "k_gemm_rows|Kernel_Name" "$f" | head -8
  grep -E "k_gemm_rows|Kernel_Name" "$f" > $OUT/pmc_${c}_pv.csv
done
rm -rf $OUT/trace/*/*.db 2>/dev/null
du -sh $OUT
