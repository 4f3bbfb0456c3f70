set -x
python -m pytest tests/test_train_backward.py tests/test_train_update.py tests/test_train_gpu.py tests/test_gemm_pkb.py tests/test_gemm_pk.py -x -q -m gpu 2>&1 | tail -8
python bench.py --train 3 --steps 8 --warmup 6 2>&1 | tail -1 > gpurun_out/pkb_train3.json
cut -c1-330 gpurun_out/pkb_train3.json
python bench.py --train 4 --steps 8 --warmup 6 2>&1 | tail -1 > gpurun_out/pkb_train4.json
cut -c1-330 gpurun_out/pkb_train4.json
