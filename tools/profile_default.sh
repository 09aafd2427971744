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
# TIMED REGION ONLY (round 6): the full kernel trace of the headline run with bench.py's marker dispatches around the timed region ->
# profiles/<tag>_timed_kernel_stats.csv (per-kernel totals of the timed launches alone: no warm-up, no graph-instantiation runs) and
# profiles/<tag>_timed_recompute.json (roofline.frac recomputed from that CSV next to the HIP-event figure of the SAME run)
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_t -o p -- python3 bench.py --steps 20 --warmup 5 --headline-only --mark-timed-region > $OUT/bench_timed_under_rocprof.json 2> $OUT/bench_timed_under_rocprof.err
python tools/timed_region_stats.py $OUT/prof_t $OUT/bench_timed_under_rocprof.json profiles/${TAG} > $OUT/timed_recompute.log 2>&1
cat $OUT/timed_recompute.log
cp profiles/${TAG}_timed_kernel_stats.csv profiles/${TAG}_timed_recompute.json $OUT/ 2>/dev/null
rm -rf $OUT/prof_t
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc/$c -o p -- python bench.py --steps 20 --warmup 5 --headline-only > $OUT/pmc_$c.log 2>&1 || echo FAILED $c
done
python tools/pmc_summary.py $OUT/pmc $TAG "bench.py --steps 20 --warmup 5 --headline-only" > $OUT/pmc_summary.log 2>&1
cat $OUT/pmc_summary.log
cp profiles/${TAG}_pmc_tp_conv_summary.csv profiles/${TAG}_traffic.json $OUT/ 2>/dev/null
rm -rf $OUT/prof $OUT/pmc
ls -la $OUT
