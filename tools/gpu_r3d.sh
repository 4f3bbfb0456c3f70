#!/usr/bin/env bash
set -u
mkdir -p gpurun_out/r3d
export TMPDIR=/tmp
O=gpurun_out/r3d
for m in 0 1 2 3 4 6 7; do echo "== CRAFT_PK_MODE=$m"; CRAFT_PK_MODE=$m timeout 300 python tools/bench_wgrad.py --cfg 3 2>/dev/null | head -3 | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['layer'], 'gemm_us', d['gemm_us'], 'pack_us', d['pack_us'])"; done 2>&1 | tee $O/pk_modes.txt
