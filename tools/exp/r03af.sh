#!/bin/bash
mkdir -p gpurun_out/$1
timeout 1500 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_op.py tests/test_gpu_finetune_loop.py -q -x -m gpu 2>&1 | tail -3
python tools/train_profile.py --batch 8 --plain 2>&1 | grep "^batch"
python tools/train_profile.py --batch 8 --rows 40 > gpurun_out/$1/train_profile_b8.txt 2>&1
grep -E "synchronisation|op calls|GPU launches|Self C" gpurun_out/$1/train_profile_b8.txt | cut -c1-250
python tools/train_bench.py --batch 32 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','forward_ms','backward_ms','batch')})"
