cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/bench_1x1.py 2>&1 | tail -5
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_hip_e2e.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -3
bash tools/gpu.sh ab
tr() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d.get('value'))"; }
for r in 1 2; do for lib in libcraft_hip_prev.so libcraft_hip.so; do
 echo "train $lib $(CRAFT_HIP_LIB=$GRAFT_REPO_ROOT/craft_amd/$lib python bench.py --train 3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tr)"
done; done
