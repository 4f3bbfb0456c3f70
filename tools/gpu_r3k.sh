#!/usr/bin/env bash
set -u
REPO=$(pwd); O=$REPO/gpurun_out/r3k; mkdir -p $O
export TMPDIR=/tmp
python tools/host_bound_train.py 2>/dev/null | tail -4 | tee $O/host_bound.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_train -o train -- python $REPO/bench.py --train 3 --steps 4 --warmup 2 --no-cpu-baseline > $O/train3_under_rocprof.json 2> $O/rocprof_train.err
python $REPO/tools/kstats.py $(find $O/trace_train -name "*kernel_stats.csv" | head -1) 45 > $O/train_cfg3_kernel_stats.txt
rm -rf $O/trace_train
cat $O/train_cfg3_kernel_stats.txt
