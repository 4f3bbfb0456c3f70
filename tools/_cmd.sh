cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_train_update.py tests/test_gemm_pkb.py tests/test_train_backward.py tests/test_trainer_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -4
timeout 600 python tools/aten_sites.py 3 > gpurun_out/aten_sites_cfg3.txt 2>&1; head -8 gpurun_out/aten_sites_cfg3.txt
tr() { python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['ms_per_step'], d.get('value'))"; }
for r in 1 2 3; do echo "train $(python bench.py --train 3 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tr)"; done
echo "train4 $(python bench.py --train 4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tr)"
