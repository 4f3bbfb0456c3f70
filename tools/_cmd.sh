cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
bash tools/gpu.sh kstats bench_kstats python $R/bench.py --no-cpu-baseline --no-train-leg --steps 10 --warmup 3 > /dev/null
bash tools/gpu.sh kstats train3_kstats python $R/bench.py --train 3 --precision mixed --steps 12 --warmup 6 --no-cpu-baseline > /dev/null
python tools/step_phases.py 3 > gpurun_out/step_phases_cfg3.txt 2>&1
python tools/bench_conv_fixed.py > gpurun_out/bench_conv_fixed.txt 2>&1; B=2 python tools/bench_conv_fixed.py >> gpurun_out/bench_conv_fixed.txt 2>&1
for h in 48 56 64; do REPS=50 H8=$h python tools/run_kernel.py gru 2>&1 | tail -1 >> gpurun_out/bench_conv_fixed.txt; done
python bench.py > gpurun_out/bench2.log 2>/dev/null
head -12 gpurun_out/bench_kstats/kernel_stats.txt; grep 'lookup\|convex\|flow_head' gpurun_out/bench_kstats/kernel_stats.txt
