cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python tools/fuzz_parity.py 60 9000 variants > gpurun_out/fuzz_parity_r4b.txt 2>&1; tail -3 gpurun_out/fuzz_parity_r4b.txt
timeout 1500 python tools/fuzz_train_parity.py 16 9100 > gpurun_out/fuzz_train_r4b.txt 2>&1; tail -3 gpurun_out/fuzz_train_r4b.txt
