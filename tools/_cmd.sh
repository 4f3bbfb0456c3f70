cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_hip_ops.py tests/test_full_size_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "attn_apply or expanded or block_height or gemm or conv" 2>&1 | tail -3
ex() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['ms_per_launch'], d['roofline']['frac'])"; }
for r in 1 2 3; do for lib in libcraft_hip_prev.so libcraft_hip.so; do
 echo "$lib $(CRAFT_HIP_LIB=$R/craft_amd/$lib python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg 2>/dev/null | ex)"
done; done
