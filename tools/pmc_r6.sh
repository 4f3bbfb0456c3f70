#!/usr/bin/env bash
# Round 6: every PMC figure bench.py quotes, re-collected on the CURRENT tree (VERDICT r5 "next" 9: no counter older than its kernel).
#   bash tools/pmc_r6.sh [groups...]   -> gpurun_out/pmc_r6/  (then: python tools/pmc_r6_json.py gpurun_out/pmc_r6 profiles/r6)
# (a) tools/pmc_r5.sh's passes (FETCH_SIZE, WRITE_SIZE, two SQ groups, kernel trace) of every hot kernel group at the bench shape,
# (b) FETCH_SIZE / WRITE_SIZE of the correlation build at configs[2] (768x1024: tools/bench_corr.py),
# (c) FETCH_SIZE / WRITE_SIZE of the weight-gradient launch the training roofline reports (tools/run_wgrad_pk.py 4 fp16).
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
O=$REPO/gpurun_out/pmc_r6; mkdir -p $O
bash $REPO/tools/pmc_r5.sh pmc_r6 "$@" > $O/groups.log 2>&1
tail -n 3 $O/groups.log
cd /tmp
for what in corr2 wgrad; do
  if [ $what = corr2 ]; then cmd="python $REPO/tools/bench_corr.py --reps 4"; else cmd="python $REPO/tools/run_wgrad_pk.py 4 fp16"; fi
  for ctr in FETCH_SIZE WRITE_SIZE; do
    d=$O/p_${what}_$ctr
    timeout 600 rocprofv3 --pmc $ctr --output-format csv -d $d -o k -- $cmd > /dev/null 2> $O/${what}_$ctr.err
    f=$(find $d -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then python $REPO/tools/kstats.py "$f" > $O/${what}_$ctr.txt; rm -f $O/${what}_$ctr.err; else tail -3 $O/${what}_$ctr.err; fi
    rm -rf $d
  done
  echo "== $what done"
done
