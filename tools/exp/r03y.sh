#!/bin/bash
mkdir -p gpurun_out/$1
timeout 1500 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_op.py tests/test_gpu_finetune_loop.py -q -x -m gpu 2>&1 | tail -8
python tools/train_profile.py --batch 8 --regions > gpurun_out/$1/regions.txt 2>&1
grep -A50 "GPU launches by" gpurun_out/$1/regions.txt
python tools/train_profile.py --batch 8 --rows 70 > gpurun_out/$1/train_profile_b8.txt 2>&1
grep -E "^batch|synchronisation|op calls|GPU launches|Self C" gpurun_out/$1/train_profile_b8.txt | cut -c1-250
