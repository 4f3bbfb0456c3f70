#!/usr/bin/env bash
set -u
mkdir -p gpurun_out/r3b
export TMPDIR=/tmp
O=gpurun_out/r3b
timeout 600 python -m pytest tests/test_gemm_pk.py -m gpu -q --tb=short -p no:cacheprovider -x > $O/pytest_pk.log 2>&1; echo "pytest pk exit: $?"; tail -n 25 $O/pytest_pk.log
timeout 300 python tools/bench_wgrad.py --cfg 3 > $O/bench_wgrad_cfg3.txt 2>&1; cat $O/bench_wgrad_cfg3.txt
timeout 900 python tools/train_parity_scan.py --iters 12 --policies fp32,train_f16x3 --f64 > $O/parity_scan_T12.txt 2>&1; cat $O/parity_scan_T12.txt
