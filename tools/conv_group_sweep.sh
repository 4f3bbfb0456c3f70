#!/usr/bin/env bash
# Sensitivity study (GPU box): flow error at 448x1024 / 12 iterations and throughput when ONE group of the update block's
# convolutions runs plain fp16 MFMA instead of f16x3 (CRAFT_CONV_GROUPS, craft_amd/hip.py).  Groups: gru, menc, fh, mask.
set -u
for g in "" "mask=fp16" "fh=fp16" "menc=fp16" "gru=fp16" "mask=fp16,fh=fp16" "mask=fp16,fh=fp16,menc=fp16"; do
  echo "== CRAFT_CONV_GROUPS='$g'"
  CRAFT_CONV_GROUPS="$g" python tools/x3_terms_eval.py mixed 2>&1 | tail -1
  CRAFT_CONV_GROUPS="$g" python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   ', d['value'], 'pairs/s', d['ms_per_step'], 'ms')"
done
