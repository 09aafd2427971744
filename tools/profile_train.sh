# Evidence for the fine-tuning step: op table + launches per region + free-running step times + rocprofv3 gap analysis.
# Usage on the GPU box: bash tools/profile_train.sh r03_z
TAG=${1:-r03_z}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
python tools/train_profile.py --batch 8 --rows 60 > $OUT/train_profile_b8.txt 2>&1
python tools/train_profile.py --batch 8 --regions > $OUT/train_regions_b8.txt 2>&1
python tools/train_profile.py --batch 8 --plain 2>&1 | grep "^batch" > $OUT/train_free_running_b8.txt
python tools/train_bench.py --batch 8 2>/dev/null | tail -1 > $OUT/train_bench_b8.json
python tools/train_bench.py --batch 32 2>/dev/null | tail -1 > $OUT/train_bench_b32.json
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o tr -- python3 tools/train_profile.py --batch 8 --plain > /dev/null 2>&1
f=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python tools/gap_stats.py $f --tail 0.5 > $OUT/train_gap_stats_under_rocprof.txt 2>&1
rm -rf $OUT/trace
cat $OUT/train_free_running_b8.txt $OUT/train_bench_b8.json
