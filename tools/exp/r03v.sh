#!/bin/bash
mkdir -p gpurun_out/$1
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$1/trace -o tr -- python3 tools/train_profile.py --batch 8 --plain > gpurun_out/$1/plain.txt 2>&1
grep "^batch" gpurun_out/$1/plain.txt
f=$(find gpurun_out/$1/trace -name "*kernel_trace.csv" | head -1)
python tools/gap_stats.py $f --tail 0.5 | tee gpurun_out/$1/gap_stats.txt
rm -rf gpurun_out/$1/trace
