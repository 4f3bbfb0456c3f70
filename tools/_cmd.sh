cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_ops.py tests/test_full_size_parity.py -m gpu -q -x --tb=short -p no:cacheprovider -k "attn or expanded or feat or flash or batch4" 2>&1 | tail -3
ex() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['frac'])"; }
for r in 1 2 3; do for lib in libcraft_hip_prev.so libcraft_hip.so; do
 echo "$lib $(CRAFT_HIP_LIB=$GRAFT_REPO_ROOT/craft_amd/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg 2>/dev/null | ex)"
done; done
for lib in libcraft_hip_prev.so libcraft_hip.so; do CRAFT_HIP_LIB=$GRAFT_REPO_ROOT/craft_amd/$lib REPS=50 python tools/run_kernel.py pv 2>&1 | tail -1; done
CRAFT_P_ROWMAJOR=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg 2>/dev/null | ex
