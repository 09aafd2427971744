#!/bin/bash
mkdir -p gpurun_out/r03_q
python tools/train_profile.py --batch 8 --rows 60 --cprofile > gpurun_out/r03_q/train_cprofile_b8.txt 2>&1
head -5 gpurun_out/r03_q/train_cprofile_b8.txt
