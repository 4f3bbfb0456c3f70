#!/usr/bin/env bash
# round 3, first GPU pass: instruction probes, the parity suite, the default bench line (with the train_cfg3 leg), training lines
set -u
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
O=gpurun_out/r3a
timeout 60 tools/ubench/tr_read.bin > $O/tr_read.txt 2>&1; echo "tr_read exit $?"
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest_gpu.log 2>&1; echo "pytest exit: $?" | tee -a $O/pytest_gpu.log
tail -n 15 $O/pytest_gpu.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; tail -n 5 $O/bench.err; cat $O/bench.json
timeout 600 python bench.py --train 3 --steps 10 --warmup 3 > $O/bench_train3.json 2> $O/bench_train3.err; echo "train3 exit $?"; tail -n 5 $O/bench_train3.err; cat $O/bench_train3.json
timeout 600 python bench.py --train 4 --steps 10 --warmup 3 > $O/bench_train4.json 2> $O/bench_train4.err; echo "train4 exit $?"; tail -n 5 $O/bench_train4.err; cat $O/bench_train4.json
grep -h "train parity" $O/pytest_gpu.log | tail -n 30
