#!/usr/bin/env bash
# One GPU-box round: parity tests, smoke, short bench.  Everything is logged under gpurun_out/.
# usage (from the build container):  gpurun --timeout 1500 -- 'bash tools/gpu_check.sh [pytest-args]'
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -m3 -E "Marketing Name|gfx" > gpurun_out/gpu.txt
nproc >> gpurun_out/gpu.txt
echo "== pytest -m gpu $*" 
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider "$@" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit: $?" | tee -a gpurun_out/pytest_gpu.log
tail -n 60 gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke exit: $?" | tee -a gpurun_out/smoke.log
tail -n 5 gpurun_out/smoke.log
echo "== bench"
timeout 600 python bench.py --steps 3 --warmup 1 --ops > gpurun_out/bench.log 2> gpurun_out/bench.err; echo "bench exit: $?"
tail -n 40 gpurun_out/bench.err; cat gpurun_out/bench.log
