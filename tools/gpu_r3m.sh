#!/usr/bin/env bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3m
python tools/step_phases.py 3 2>/dev/null | tee gpurun_out/r3m/step_phases_cfg3.txt
