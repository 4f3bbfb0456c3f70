#!/usr/bin/env bash
# GPU idle-gap analysis of the bench step: kernel-trace of bench.py, then per-step busy time vs wall span.
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/tr_bench
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_bench -o k -- python $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline $BENCH_ARGS > /dev/null 2>&1
f=$(find /tmp/tr_bench -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find step boundaries: k_stem7x7 with the fnet grid marks the start of a forward (two stems per forward: fnet first)
starts = [i for i, r in enumerate(rows) if "k_stem" in r["Kernel_Name"]]
starts = starts[0::2]
print("forwards found:", len(starts))
for a, b in zip(starts[-4:-1], starts[-3:]):
    seg = rows[a:b]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    gaps = []
    for x, y in zip(seg[:-1], seg[1:]):
        g = int(y["Start_Timestamp"]) - int(x["End_Timestamp"])
        gaps.append((g, re.sub(r"\(.*", "", x["Kernel_Name"])[:40], re.sub(r"\(.*", "", y["Kernel_Name"])[:40]))
    print(f"step span {(t1 - t0) / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  kernels {len(seg)}  idle {(t1 - t0 - busy) / 1e6:.3f} ms")
    gaps.sort(reverse=True)
    for g, x, y in gaps[:6]:
        print(f"    gap {g / 1e3:8.1f} us after {x} before {y}")
PY
