#!/bin/bash
mkdir -p gpurun_out/$1
python tools/train_profile.py --batch 8 --regions > gpurun_out/$1/regions.txt 2>&1
grep -A70 "GPU launches by" gpurun_out/$1/regions.txt
