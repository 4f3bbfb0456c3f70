cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python tools/train_soak.py mixed 300 > gpurun_out/train_soak_r4b.txt 2>&1; tail -16 gpurun_out/train_soak_r4b.txt
