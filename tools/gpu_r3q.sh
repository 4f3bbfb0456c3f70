#!/usr/bin/env bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r3q
for w in 8 12; do timeout 900 python tools/feed_bench.py --workers $w > gpurun_out/r3q/feed_bench_w$w.json 2> gpurun_out/r3q/feed_bench.err; cat gpurun_out/r3q/feed_bench_w$w.json; done
nproc
