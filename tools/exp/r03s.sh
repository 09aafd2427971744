#!/bin/bash
mkdir -p gpurun_out/r03_s
timeout 1500 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_op.py tests/test_gpu_finetune_loop.py -q -x -m gpu 2>&1 | tail -3
python tools/train_profile.py --batch 8 --rows 60 --cprofile > gpurun_out/r03_s/train_cprofile_b8.txt 2>&1
head -5 gpurun_out/r03_s/train_cprofile_b8.txt
python tools/train_bench.py --batch 8 2>&1 | tail -1 | cut -c1-100,600-900
