#!/usr/bin/env bash
# Package power and shader clock while each hot kernel group of tools/run_kernel.py runs back to back for a few seconds (rocm-smi sampled
# from the shell; samples below 600 W -- the process starting up -- dropped).  usage: bash tools/power_by_kernel.sh  -> stdout
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
run() {  # group reps
  REPS=$2 python $REPO/tools/run_kernel.py $1 mixed > /tmp/pbk_$1.out 2>&1 &
  pid=$!
  s=""
  while kill -0 $pid 2>/dev/null; do
    r=$(rocm-smi --showclocks --showpower 2>/dev/null | grep "GPU\[0\]")
    p=$(echo "$r" | grep "Power" | sed 's/.*: //')
    c=$(echo "$r" | grep "sclk" | sed 's/.*(//; s/Mhz)//')
    if [ "${p%.*}" -gt 600 ] 2>/dev/null; then s="$s $c/${p%.*}"; fi
    sleep 0.25
  done
  echo "$1 (REPS=$2): sclk MHz / package W:$s"
  grep -h "us per launch\|us per" /tmp/pbk_$1.out | head -2
}
run pv 9000
run grustep 9000
run convtok 30000
run menc 9000
run flash 3500
run corr 3500
run probs 5000
run fnet 1200
