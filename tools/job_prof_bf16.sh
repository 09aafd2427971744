# kernel statistics of the C4 bf16 run under the three role modes:   gpurun -- bash tools/job_prof_bf16.sh [modes]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
MODES=${1:-"0 2"}
for r in $MODES; do
  OUT=gpurun_out/prof_bf16_r$r
  rm -rf $OUT; mkdir -p $OUT
  export CBD_BF16_ROLES=$r
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python3 bench.py --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps 2 --warmup 1 --headline-only --diag-library --pair 2 > $OUT/log.txt 2>&1
  find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
  rm -rf $OUT/prof
  echo "== roles $r"; head -8 $OUT/kernel_stats.csv | cut -c1-200; tail -1 $OUT/log.txt | cut -c1-160
done
