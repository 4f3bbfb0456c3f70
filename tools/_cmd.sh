cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gemm_pkb.py tests/test_train_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "column_block or flow_tokens or convex_upsample" 2>&1 | tail -15
