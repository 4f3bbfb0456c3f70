#!/usr/bin/env bash
# same-box A/B of two builds: craft_amd/libcraft_hip.so vs craft_amd/libcraft_hip_prev.so (copy the old build there first)
for rep in 1 2; do
for lib in libcraft_hip_prev.so libcraft_hip.so; do
  CRAFT_HIP_LIB=$(pwd)/craft_amd/$lib python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['value'], d['ms_per_step'])"
done; done
