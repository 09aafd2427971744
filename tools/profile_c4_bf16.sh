# rocprofv3 kernel stats of the C4-as-specified bench (64 samples x 40 steps, bf16): bash tools/profile_c4_bf16.sh r02_m
set -x
TAG=${1:-r02_m}
PAIR=${2:-8}      # complexes per launch: the c4_bf16 leg of bench.py co-schedules eight since round 6 (two in round 5)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python bench.py --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps 8 --warmup 2 --pair $PAIR --headline-only --no-cpu-baseline > $OUT/bench_c4_bf16_under_rocprof.json 2> $OUT/bench_c4_bf16.err
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/c4_bf16_kernel_stats.csv \;
rm -rf $OUT/prof
head -5 $OUT/c4_bf16_kernel_stats.csv
