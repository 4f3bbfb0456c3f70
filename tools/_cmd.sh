cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_train_gpu.py tests/test_hip_ops.py tests/test_train_update.py tests/test_hip_e2e.py -m gpu -q --tb=short -p no:cacheprovider -k "convex or upsample or golden or fused_iteration or e2e" 2>&1 | tail -3
bash tools/gpu.sh kstats cu python $R/bench.py --train 3 --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | grep 'convex'
CRAFT_HIP_LIB=$R/craft_amd/libcraft_hip_prev.so bash tools/gpu.sh kstats cu0 python $R/bench.py --train 3 --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | grep 'convex'
