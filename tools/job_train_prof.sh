TAG=${1:-r04_b}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
python tools/train_bench.py --batch 8 2>/dev/null | tail -1 > $OUT/train_bench_b8.json
python tools/train_bench.py --batch 32 2>/dev/null | tail -1 > $OUT/train_bench_b32.json
for b in 8 32; do
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof$b -o p -- python3 tools/train_profile.py --batch $b --plain > $OUT/train_plain_b$b.txt 2>&1
find $OUT/prof$b -name "*kernel_stats.csv" -exec cp {} $OUT/train_kernel_stats_b$b.csv \;
f=$(find $OUT/prof$b -name "*kernel_trace.csv" | head -1)
python tools/gap_stats.py $f --tail 0.5 > $OUT/train_gap_stats_b$b.txt 2>&1
rm -rf $OUT/prof$b
done
python tools/train_profile.py --batch 8 --rows 50 > $OUT/train_profile_b8.txt 2>&1
grep "^batch" $OUT/train_plain_b8.txt $OUT/train_plain_b32.txt; cut -c 1-700 $OUT/train_bench_b8.json; echo; cut -c 1-700 $OUT/train_bench_b32.json; echo; head -12 $OUT/train_gap_stats_b8.txt
