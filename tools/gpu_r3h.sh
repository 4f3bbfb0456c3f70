#!/usr/bin/env bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3h
timeout 300 python tools/debug_train_steps.py auto 6 2>&1 | grep -v Warn | tee gpurun_out/r3h/steps_auto.txt
timeout 900 python tools/train_parity_scan.py --iters 12 --policies train_f16x3 > gpurun_out/r3h/parity_scan_T12.txt 2>&1; grep -v Warn gpurun_out/r3h/parity_scan_T12.txt
timeout 600 python -m pytest tests/test_train_backward.py -m gpu -q --tb=short -p no:cacheprovider -x -k "configs3_size or configs4_shape" 2>&1 | tail -5
