#!/usr/bin/env bash
set -u
mkdir -p gpurun_out/r3e
export TMPDIR=/tmp
O=gpurun_out/r3e
timeout 600 python -m pytest tests/test_gemm_pk.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest_pk.log 2>&1; echo "pytest pk exit: $?"; tail -n 5 $O/pytest_pk.log
timeout 300 python tools/bench_wgrad.py --cfg 3 > $O/bench_wgrad_cfg3.txt 2>&1; cat $O/bench_wgrad_cfg3.txt
timeout 900 python tools/train_parity_scan.py --iters 12 --policies train_f16x3 > $O/parity_scan_T12.txt 2>&1; grep -v Warn $O/parity_scan_T12.txt
timeout 1500 python -m pytest tests/test_train_backward.py tests/test_trainer_gpu.py tests/test_train_encoder.py tests/test_fuzz_parity.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest_train.log 2>&1; echo "pytest train exit: $?"; tail -n 15 $O/pytest_train.log
timeout 600 python bench.py --train 3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train3.json 2> $O/bench_train3.err; echo "train3 exit $?"; tail -n 3 $O/bench_train3.err; cat $O/bench_train3.json
CRAFT_NO_PK=1 timeout 600 python bench.py --train 3 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train3_nopk.json 2> $O/bench_train3_nopk.err; echo "train3 nopk exit $?"; cat $O/bench_train3_nopk.json
