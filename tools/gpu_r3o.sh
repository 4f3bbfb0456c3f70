#!/usr/bin/env bash
export TMPDIR=/tmp
REPO=$(pwd); O=$REPO/gpurun_out/r3o; mkdir -p $O
timeout 1500 python -m pytest tests/test_train_backward.py tests/test_train_update.py tests/test_trainer_gpu.py -m gpu -q --tb=short -p no:cacheprovider -s 2>&1 | grep -E "train parity|fused update|passed|failed|Error|assert" | tail -60 | tee $O/pytest_train.txt
