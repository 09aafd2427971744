#!/bin/bash
mkdir -p gpurun_out/r03_p
python tools/train_profile.py --batch 8 --rows 70 > gpurun_out/r03_p/train_profile_b8.txt 2>&1
head -10 gpurun_out/r03_p/train_profile_b8.txt | cut -c1-250
python tools/train_bench.py --batch 8 2>&1 | tail -1
python tools/train_bench.py --batch 32 2>&1 | tail -1
