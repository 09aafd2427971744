#!/bin/bash
timeout 1500 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_op.py tests/test_gpu_finetune_loop.py -q -x -m gpu 2>&1 | tail -3
for rep in 1 2; do
  python tools/train_profile.py --batch 8 --plain 2>&1 | grep "^batch"
done
python tools/train_bench.py --batch 8 2>&1 | tail -1 | cut -c1-60,560-900
