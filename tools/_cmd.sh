cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_ops.py tests/test_train_backward.py tests/test_hip_e2e.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm or linear or golden or layouts or expanded or e2e" 2>&1 | tail -3
python tools/bench_1x1.py 2>&1 | tail -5
bash tools/gpu.sh ab
tr() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d.get('value'))"; }
for r in 1 2; do for lib in libcraft_hip_prev.so libcraft_hip.so; do
 echo "train $lib $(CRAFT_HIP_LIB=$R/craft_amd/$lib python bench.py --train 3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tr)"
done; done
