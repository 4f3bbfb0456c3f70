cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_train_backward.py tests/test_train_update.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -4
tr() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d.get('value'))"; }
for r in 1 2 3; do for lib in libcraft_hip_prev.so libcraft_hip.so; do
 echo "train $lib $(CRAFT_HIP_LIB=$R/craft_amd/$lib python bench.py --train 3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tr)"
done; done
bash tools/gpu.sh kstats lkb python $R/bench.py --train 3 --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | grep 'lookup'
