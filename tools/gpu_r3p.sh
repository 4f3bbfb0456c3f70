#!/usr/bin/env bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3p
timeout 600 python tools/p_fp8_eval.py 2>&1 | grep -v Warn | tee gpurun_out/r3p/p_fp8_eval.txt
