# evidence after the bf16 default changed (register-stationary kernel): full -m gpu log, default bench lines, C4 kernel stats + PMC passes
TAG=${1:-r05_zz}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_lines.json 2> $OUT/bench_line.err; tail -c 700 $OUT/bench_lines.json
bash tools/profile_c4_bf16.sh $TAG > $OUT/profile_c4.log 2>&1
bash tools/pmc_c4_bf16.sh $TAG 2 > $OUT/pmc_c4.log 2>&1
tail -3 $OUT/pmc_c4.log | cut -c1-600
ls -la $OUT
