cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
bash tools/gpu.sh kstats bench_kstats python $R/bench.py --no-cpu-baseline --no-train-leg --steps 10 --warmup 3 > /dev/null
bash tools/gpu.sh kstats train3_kstats python $R/bench.py --train 3 --precision mixed --steps 12 --warmup 6 --no-cpu-baseline > /dev/null
python tools/step_phases.py 3 > gpurun_out/step_phases_cfg3.txt 2>&1
head -3 gpurun_out/bench_kstats/kernel_stats.txt; head -3 gpurun_out/train3_kstats/kernel_stats.txt
