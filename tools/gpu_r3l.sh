#!/usr/bin/env bash
set -u
REPO=$(pwd); O=$REPO/gpurun_out/r3l; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_train -o train -- python $REPO/bench.py --train 3 --steps 3 --warmup 2 --no-cpu-baseline > $O/train3_under_rocprof.json 2> $O/rocprof_train.err
python $REPO/tools/trace_gaps.py $(find $O/trace_train -name "*kernel_trace.csv" | head -1) 240 > $O/train_gaps.txt
rm -rf $O/trace_train
cat $O/train_gaps.txt
