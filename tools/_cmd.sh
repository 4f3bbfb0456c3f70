cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_train_backward.py -m gpu -q --tb=short -p no:cacheprovider -k "corr or volume or training_step" 2>&1 | tail -3
for lib in libcraft_hip_prev.so libcraft_hip.so; do CRAFT_HIP_LIB=$R/craft_amd/$lib bash tools/gpu.sh kstats cp_$lib python $R/bench.py --train 3 --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | grep "corr_p"; done
