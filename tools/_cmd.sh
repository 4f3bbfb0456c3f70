cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/aten_sites.py 3 > gpurun_out/aten_sites_cfg3.txt 2>&1; head -60 gpurun_out/aten_sites_cfg3.txt
timeout 1500 python -m pytest tests/test_train_update.py tests/test_trainer_gpu.py tests/test_train_gpu.py tests/test_train_backward.py tests/test_reference_wrappers.py tests/test_bench_contract.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -8
tr() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d.get('value'))"; }
for r in 1 2; do
 echo "direct   $(python bench.py --train 3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tr)"
 echo "backward $(CRAFT_TRAINER_BACKWARD=1 python bench.py --train 3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tr)"
done
