# round 6, job b: new tests (bench ranks on a shared GPU, pipelined set-up after the ADVICE fixes, training ops after the split), the
# default bench with the new legs, the timed-region-only kernel summary
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_b
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_bench_ranks.py tests/test_gpu_finetune_loop.py tests/test_gpu_train_op.py tests/test_gpu_train_graph.py tests/test_gpu_complex_set.py tests/test_gpu_bf16.py -q -x > $OUT/pytest_new.log 2>&1; tail -5 $OUT/pytest_new.log
timeout 1500 python bench.py --steps 20 --warmup 5 > $OUT/bench_lines.json 2> $OUT/bench.err; tail -c 1800 $OUT/bench_lines.json; tail -5 $OUT/bench.err
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_t -o p -- python3 bench.py --steps 20 --warmup 5 --headline-only --mark-timed-region > $OUT/bench_timed_under_rocprof.json 2> $OUT/bench_timed_under_rocprof.err
python tools/timed_region_stats.py $OUT/prof_t $OUT/bench_timed_under_rocprof.json $OUT/r06_b > $OUT/timed_recompute.log 2>&1
cat $OUT/timed_recompute.log
rm -rf $OUT/prof_t
ls -la $OUT
