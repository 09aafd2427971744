# rocprofv3 evidence for the default bench command (driver's flags): kernel stats + the four PMC passes (separate runs, no tracing with
# --pmc), summarised into profiles/<tag>_*.  Usage on the GPU box: bash tools/profile_default.sh r02_e
set -x
TAG=${1:-r02_e}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python bench.py --steps 20 --warmup 5 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/prof
# the same with the headline region alone: the average tp_conv<3,3> duration of this file is the one bench.py's HIP events must agree with
# (the full command above also runs the other legs' launches of the same kernel on other workloads)
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python bench.py --steps 20 --warmup 5 --headline-only > $OUT/bench_headline_under_rocprof.json 2> $OUT/bench_headline_under_rocprof.err
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/headline_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc/$c -o p -- python bench.py --steps 20 --warmup 5 --headline-only > $OUT/pmc_$c.log 2>&1 || echo FAILED $c
done
python tools/pmc_summary.py $OUT/pmc $TAG "bench.py --steps 20 --warmup 5 --headline-only" > $OUT/pmc_summary.log 2>&1
cat $OUT/pmc_summary.log
cp profiles/${TAG}_pmc_tp_conv_summary.csv profiles/${TAG}_traffic.json $OUT/ 2>/dev/null
rm -rf $OUT/prof $OUT/pmc
ls -la $OUT
