cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_tg
rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python tools/train_graph_check.py --batch 8 --skip-checks --modes hip_graph > $OUT/log.txt 2>&1
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
python tools/gap_stats.py $OUT/kernel_trace.csv > $OUT/gaps.txt 2>&1
head -45 $OUT/gaps.txt; tail -2 $OUT/log.txt | cut -c 1-600
rm -rf $OUT/prof; gzip -f $OUT/kernel_trace.csv
