set -u
timeout 1200 python -m pytest tests/test_train_backward.py tests/test_trainer_gpu.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -4
for rep in 1 2; do
  python bench.py --train 3 --steps 10 --warmup 3 --torch-encoders 2>&1 | tail -1 | cut -c100-330
  python bench.py --train 3 --steps 10 --warmup 3 2>&1 | tail -1 | cut -c100-330
done
python bench.py --train 4 --steps 10 --warmup 3 --torch-encoders 2>&1 | tail -1 | cut -c100-330
python bench.py --train 4 --steps 10 --warmup 3 2>&1 | tail -1 | cut -c100-330
python tools/train_breakdown.py 3 2>&1 | tail -8
