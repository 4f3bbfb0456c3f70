#!/usr/bin/env bash
# HBM write traffic of the fused correlation build (k_corr_build4t) by pyramid level: rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes
# of tools/bench_corr.py with the stores of level 0 / levels 1-3 compiled in or masked (CRAFT_CORR_DBG), plus the kernel times.
set -u
REPO=$(pwd); O=$REPO/gpurun_out/corr_pmc; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for dbg in 0 1 2 3; do
  for c in WRITE_SIZE FETCH_SIZE; do
    CRAFT_CORR_DBG=$dbg timeout 300 rocprofv3 --pmc $c --output-format csv -d $O/p_${dbg}_$c -o k -- python $REPO/tools/bench_corr.py --reps 4 > /dev/null 2> $O/err.txt
    f=$(find $O/p_${dbg}_$c -name "*counter_collection.csv" | head -1)
    python - "$f" $dbg $c <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_corr_build4t" in r["Kernel_Name"]]
v = [float(r["Counter_Value"]) for r in rows]
print(f"CRAFT_CORR_DBG={sys.argv[2]} {sys.argv[3]:10s} launches {len(v):2d} mean {sum(v) / max(1, len(v)):14.1f} (raw counter units: KB)")
PY
    rm -rf $O/p_${dbg}_$c
  done
  CRAFT_CORR_DBG=$dbg python $REPO/tools/bench_corr.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   un-profiled build ms', d['corr_build']['ms'], 'pyramid bytes', d['volume_bytes'])"
done
