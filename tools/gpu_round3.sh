#!/usr/bin/env bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider "$@" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit: $?" | tee -a gpurun_out/pytest_gpu.log
tail -n 40 gpurun_out/pytest_gpu.log | cut -c1-300
timeout 600 python tools/precision_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/precision_sweep.log
timeout 600 python bench.py --ops --no-cpu-baseline > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit $?"; grep "\[ops\]" gpurun_out/bench.err | head -12; cut -c1-700 gpurun_out/bench.log
