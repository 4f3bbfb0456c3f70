set -u
timeout 1200 python -m pytest tests/test_train_backward.py tests/test_trainer_gpu.py tests/test_train_gpu.py tests/test_train_encoder.py -x -q -m gpu 2>&1 | tail -2
python bench.py --train 3 --steps 10 --warmup 3 2>&1 | tail -1 | cut -c100-260
python bench.py --train 4 --steps 10 --warmup 3 2>&1 | tail -1 | cut -c100-260
python tools/host_bound_train.py 3 | tail -1
