#!/usr/bin/env bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3i
timeout 900 python -m pytest tests/test_train_update.py -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -v Warn | tail -40 | tee gpurun_out/r3i/pytest_update.txt
