set -x
python -m pytest tests/test_train_backward.py -x -q -m gpu 2>&1 | tail -5
python -m pytest tests/test_train_update.py tests/test_train_gpu.py -x -q -m gpu 2>&1 | tail -3
python bench.py --train 3 --steps 8 --warmup 6 2>&1 | tail -1 > gpurun_out/drop_train3.json
cat gpurun_out/drop_train3.json | cut -c1-400
