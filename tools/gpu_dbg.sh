#!/usr/bin/env bash
# kernel-time A/B over CRAFT_DBG variants of the GRU conv (diagnosis only)
REPO=$(pwd); export TMPDIR=/tmp; cd /tmp
for d in 0 1 2 3 4 5 6 7 8; do
  rm -rf /tmp/dbg$d
  CRAFT_DBG=$d rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dbg$d -o k -- python $REPO/tools/run_kernel.py gru mixed > /dev/null 2>&1
  echo "DBG=$d"; python $REPO/tools/kstats.py $(find /tmp/dbg$d -name "*kernel_stats.csv" | head -1) 3 | grep conv_halo
done
