cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_e2e.py tests/test_train_encoder.py -m gpu -q --tb=short -p no:cacheprovider -k "stem or encoder or golden or e2e" 2>&1 | tail -3
