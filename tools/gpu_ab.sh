#!/usr/bin/env bash
# A/B of environment-variable tuning knobs on the bench.  usage: bash tools/gpu_ab.sh VAR v1 v2 ...
set -u; export TMPDIR=/tmp; mkdir -p gpurun_out
VAR=$1; shift
for v in "$@"; do
  echo "== $VAR=$v"
  env $VAR=$v timeout 300 python bench.py --ops --no-cpu-baseline --steps 6 --warmup 2 2>&1 >/dev/null | grep "\[ops\]" | head -7
  env $VAR=$v timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pairs/s', d['value'], 'ms/step', d['ms_per_step'], 'conv TF', d['roofline']['achieved'])"
done
