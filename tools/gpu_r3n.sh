#!/usr/bin/env bash
set -u
REPO=$(pwd); O=$REPO/gpurun_out/r3n; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; echo "pytest exit: $?"; tail -n 4 $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_conv']['ms_per_launch']); print(json.dumps(d['train_cfg3'])[:900])"
timeout 600 python bench.py --train 3 --steps 10 --warmup 6 > $O/bench_train3.json 2> $O/bench_train3.err; echo "train3 exit $?"; cut -c1-260 $O/bench_train3.json
timeout 600 python bench.py --train 4 --steps 10 --warmup 6 > $O/bench_train4.json 2> $O/bench_train4.err; echo "train4 exit $?"; cut -c1-260 $O/bench_train4.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_train -o train -- python $REPO/bench.py --train 3 --steps 4 --warmup 2 --no-cpu-baseline > $O/train3_under_rocprof.json 2> $O/rocprof_train.err
python $REPO/tools/kstats.py $(find $O/trace_train -name "*kernel_stats.csv" | head -1) 50 > $O/train_cfg3_kernel_stats.txt
rm -rf $O/trace_train
cat $O/train_cfg3_kernel_stats.txt
