#!/usr/bin/env bash
# per-dispatch kernel trace of one tools/run_kernel.py mode (last repetition).  usage: gpu_trace.sh <which> [policy]
W=${1:-fnet}; POL=${2:-mixed}
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/tr_$W
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$W -o k -- python $REPO/tools/run_kernel.py $W $POL > /dev/null 2>&1
f=$(find /tmp/tr_$W -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]          # second repetition
t0 = int(rows[0]["Start_Timestamp"])
tot = 0
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("craft::", "")[:46]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {d:8.1f} us  grid {r.get('Grid_Size_X', '?'):>8s} x {r.get('Grid_Size_Y', '?'):>4s}  {name}")
print(f"sum of kernel durations {tot / 1e3:.3f} ms; span {(int(rows[-1]['End_Timestamp']) - t0) / 1e6:.3f} ms")
PY
