cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_e2e.py tests/test_train_update.py -m gpu -q --tb=short -p no:cacheprovider -k "motion or convf1 or golden or update or fused_iteration or e2e" 2>&1 | tail -2
for lib in libcraft_hip_prev.so libcraft_hip.so; do CRAFT_HIP_LIB=$R/craft_amd/$lib bash tools/gpu.sh kstats f1_$lib python $R/bench.py --no-cpu-baseline --no-train-leg --steps 5 --warmup 2 2>/dev/null | grep 'convf1'; done
bash tools/gpu.sh ab
