#!/usr/bin/env bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?" | tee -a gpurun_out/pytest_gpu.log
tail -n 30 gpurun_out/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?"; cat gpurun_out/bench.log
bash tools/gpu_profile.sh r1
