# kernel-trace stats of the training bench (configs[3] by default): full kernel names
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
O=$REPO/gpurun_out/kstats_train
mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o k -- python $REPO/bench.py --train ${1:-3} --steps 10 --warmup 6 --no-cpu-baseline --precision ${2:-mixed} > $O/bench.txt 2>&1
f=$(find $O/kt -name "*kernel_stats.csv" | head -1)
cp "$f" $O/kernel_stats_full.csv
python $REPO/tools/kstats.py "$f" 70 > $O/kernel_stats.txt
rm -rf $O/kt
head -75 $O/kernel_stats.txt
tail -1 $O/bench.txt | cut -c1-300
