#!/usr/bin/env bash
# Same-box A/B of an environment knob: bench.py without / with "$1" (e.g. CRAFT_MENC_FORK=1), alternating, 3 rounds each.
# usage: gpurun -- 'bash tools/gpu_env_ab.sh CRAFT_MENC_FORK=1 [bench args]'
KNOB=$1; shift
for r in 1 2 3; do
  a=$(python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  b=$(env $KNOB python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
  echo "round $r: base $a   $KNOB $b"
done
