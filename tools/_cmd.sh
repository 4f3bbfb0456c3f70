cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_ops.py tests/test_hip_e2e.py tests/test_train_encoder.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -3
bash tools/gpu.sh ab
python tools/bench_conv_fixed.py 2>&1 | tail -4
CRAFT_HIP_LIB=$R/craft_amd/libcraft_hip_prev.so python tools/bench_conv_fixed.py 2>&1 | tail -4
