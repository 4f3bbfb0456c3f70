cd $GRAFT_REPO_ROOT
timeout 1700 python -m pytest tests/test_bench_contract.py -m gpu -q --tb=short -p no:cacheprovider -k "two_rank_default or rccl_world1" 2>&1 | tail -15
