cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fuzz_parity.py tests/test_train_encoder.py tests/test_full_size_parity.py tests/test_hip_ops.py tests/test_hip_e2e.py tests/test_train_update.py tests/test_trainer_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -8
