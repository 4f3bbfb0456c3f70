#!/usr/bin/env bash
set -u
mkdir -p gpurun_out/r3f
export TMPDIR=/tmp
timeout 600 python tools/grad_range.py 2>/dev/null | tee gpurun_out/r3f/grad_range_cfg3.txt
timeout 600 python tools/grad_range.py --B 4 --W 768 --seed 7 2>/dev/null | tee gpurun_out/r3f/grad_range_cfg4.txt
