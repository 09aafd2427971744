#!/bin/bash
mkdir -p gpurun_out/$1
python tools/train_profile.py --batch 8 --rows 70 > gpurun_out/$1/train_profile_b8.txt 2>&1
grep -E "^batch|synchronisation|op calls|GPU launches|Self C" gpurun_out/$1/train_profile_b8.txt | cut -c1-250
