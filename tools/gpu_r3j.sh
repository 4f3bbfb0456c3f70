#!/usr/bin/env bash
set -u
REPO=$(pwd); O=$REPO/gpurun_out/r3j; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_train_backward.py tests/test_trainer_gpu.py tests/test_train_encoder.py tests/test_fuzz_parity.py tests/test_gemm_pk.py tests/test_train_update.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest_train.log 2>&1; echo "pytest train exit: $?"; tail -n 4 $O/pytest_train.log
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_train -o train -- python $REPO/bench.py --train 3 --steps 4 --warmup 2 --no-cpu-baseline > $O/train3_under_rocprof.json 2> $O/rocprof_train.err
python $REPO/tools/kstats.py $(find $O/trace_train -name "*kernel_stats.csv" | head -1) 60 > $O/train_cfg3_kernel_stats.txt
rm -rf $O/trace_train
cat $O/train_cfg3_kernel_stats.txt
cd $REPO
python tools/host_bound_train.py 2>/dev/null | tail -5
