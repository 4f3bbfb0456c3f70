cd $GRAFT_REPO_ROOT; python tools/bench_conv_fixed.py 2>&1 | tail -4; B=2 python tools/bench_conv_fixed.py 2>&1 | tail -4
