# round 6, job y: evidence for the C4 bf16 leg after the cost-weighted work split (kernel stats, PMC passes, traffic), bf16 tests + soak,
# and the full bench on the final tree
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=r06_y
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_configs.py tests/test_gpu_variants.py -q > $OUT/pytest_bf16.log 2>&1; tail -2 $OUT/pytest_bf16.log
timeout 900 python tools/bf16_repeat.py 6 bf16 2>&1 | tail -1 | tee $OUT/bf16_repeat.txt
bash tools/profile_c4_bf16.sh $TAG > $OUT/profile_c4.log 2>&1
bash tools/pmc_c4_bf16.sh $TAG 2 > $OUT/pmc_c4.log 2>&1; tail -2 $OUT/pmc_bf16_summary.log | cut -c1-600
timeout 1800 python bench.py --steps 20 --warmup 5 > $OUT/bench_lines.json 2> $OUT/bench.err; tail -c 1700 $OUT/bench_lines.json
