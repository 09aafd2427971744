#!/bin/bash
mkdir -p gpurun_out/$1
python tools/train_profile.py --batch 8 --rows 400 --cprofile > gpurun_out/$1/cprofile.txt 2>&1
grep "^batch" gpurun_out/$1/cprofile.txt
