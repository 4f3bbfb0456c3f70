#!/usr/bin/env bash
# training-step state after the packed weight gradients: tests, bench line, kernel profile
set -u
REPO=$(pwd); O=$REPO/gpurun_out/r3g; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_train_backward.py tests/test_trainer_gpu.py tests/test_train_encoder.py tests/test_fuzz_parity.py tests/test_gemm_pk.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest_train.log 2>&1; echo "pytest train exit: $?"; tail -n 8 $O/pytest_train.log
timeout 600 python bench.py --train 3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train3.json 2> $O/bench_train3.err; echo "train3 exit $?"; tail -n 3 $O/bench_train3.err; cat $O/bench_train3.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_train -o train -- python $REPO/bench.py --train 3 --steps 4 --warmup 2 --no-cpu-baseline > $O/train3_under_rocprof.json 2> $O/rocprof_train.err
python $REPO/tools/kstats.py $(find $O/trace_train -name "*kernel_stats.csv" | head -1) 70 > $O/train_cfg3_kernel_stats.txt
rm -rf $O/trace_train
cat $O/train_cfg3_kernel_stats.txt
