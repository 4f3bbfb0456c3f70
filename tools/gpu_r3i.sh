#!/usr/bin/env bash
export TMPDIR=/tmp
REPO=$(pwd); O=$REPO/gpurun_out/r3i; mkdir -p $O
timeout 900 python -m pytest tests/test_train_update.py tests/test_gemm_pk.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | grep -v Warn | tail -4 | tee $O/pytest_update.txt
timeout 600 python bench.py --train 3 --steps 10 --warmup 6 --no-cpu-baseline 2>/dev/null | cut -c1-330 | tee $O/bench_train3.json
python tools/step_phases.py 3 2>/dev/null | grep "step 5\|Trainer.step"
