cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_integration_doc.py -m gpu -q --tb=short -p no:cacheprovider -k "radii or doc_stub" 2>&1 | tail -12
