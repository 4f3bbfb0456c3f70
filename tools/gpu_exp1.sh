#!/usr/bin/env bash
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
echo "== cpu baseline thread scan"
for t in 16 32 64 128; do timeout 200 python tools/cpu_baseline.py --threads $t 2>&1 | tail -1; done | tee gpurun_out/cpu_scan.log
echo "== precision sweep"
timeout 900 python tools/precision_sweep.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/precision_sweep.log
echo "== bench fp32 / bf16"
timeout 300 python bench.py --steps 3 --warmup 1 --ops --precision fp32 --no-cpu-baseline > gpurun_out/bench_fp32.log 2> gpurun_out/bench_fp32.err; tail -25 gpurun_out/bench_fp32.err; cat gpurun_out/bench_fp32.log
timeout 300 python bench.py --steps 5 --warmup 2 --precision bf16 --no-cpu-baseline > gpurun_out/bench_bf16.log 2> gpurun_out/bench_bf16.err; cat gpurun_out/bench_bf16.log
