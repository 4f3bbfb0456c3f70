#!/usr/bin/env bash
# bench.py at the other BASELINE shapes / batch sizes (5 steps each): one "HxW Bn Tt: pairs/s ms" line per shape.
run() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$*', '->', d['value'], 'pairs/s', d['ms_per_step'], 'ms/step')"; }
run --batch 1 --height 128 --width 256 --iters 4
run --batch 1
run --batch 2
run --batch 8
run --batch 1 --height 768 --width 1024
run --batch 8 --height 368 --width 496
run --batch 4 --height 368 --width 768
run --batch 4 --height 376 --width 1248
python tools/bench_corr.py 2>/dev/null | tail -1 | cut -c1-600
