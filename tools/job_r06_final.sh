# round 6, final: the complete -m gpu suite, the experiments' test, smoke and the default bench (driver's flags) on the final tree
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_final2
mkdir -p $OUT
timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 600 python -m pytest experiments/test_role_split.py -q > $OUT/pytest_experiments.log 2>&1; tail -1 $OUT/pytest_experiments.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -3 $OUT/smoke.log
timeout 1800 python bench.py --steps 20 --warmup 5 > $OUT/bench_lines.json 2> $OUT/bench.err; tail -c 1700 $OUT/bench_lines.json
