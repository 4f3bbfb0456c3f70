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
python $REPO/tools/kstats.py $OUT/kernel_stats.csv 22
for k in pv gru; do
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_${k}_$c -o k -- python $REPO/tools/run_kernel.py $k mixed > /dev/null 2> $OUT/pmc_${k}_$c.err
  f=$(find $OUT/pmc_${k}_$c -name "*counter_collection.csv" | head -1)
  echo "== $k $c"; python $REPO/tools/kstats.py "$f" | grep -A2 -E "k_pv16|k_conv_halo"
  grep -E "k_pv16|k_conv_halo|Kernel_Name" "$f" > $OUT/pmc_${k}_${c}.csv
done; done
rm -rf $OUT/trace/*/*.db 2>/dev/null
du -sh $OUT
