# GPU busy share and idle gaps of the heterogeneous complex-set run (configs[2] in small)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/set_prof
rm -rf $OUT; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python3 tools/run_set.py --complexes 24 > $OUT/log.txt 2>&1
f=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/gap_stats.py $f --tail 0.8 --after-largest-gap > $OUT/gaps.txt 2>&1
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/prof
tail -2 $OUT/log.txt | cut -c1-900; head -64 $OUT/gaps.txt
