cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_train_encoder.py tests/test_train_backward.py -m gpu -q --tb=short -p no:cacheprovider -k "encoder or norm or training_step" 2>&1 | tail -3
for lib in libcraft_hip_prev.so libcraft_hip.so; do CRAFT_HIP_LIB=$R/craft_amd/$lib bash tools/gpu.sh kstats na_$lib python $R/bench.py --train 3 --steps 6 --warmup 3 --no-cpu-baseline 2>/dev/null | grep 'norm_act'; done
tr() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d.get('value'))"; }
for r in 1 2; do for lib in libcraft_hip_prev.so libcraft_hip.so; do
 echo "train $lib $(CRAFT_HIP_LIB=$R/craft_amd/$lib python bench.py --train 3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tr)"
done; done
