# round 6, job e: the full bench on the final tree (side stream picked by probing for a free hardware queue; cpu_baseline over all steps),
# engine memory, the two-rank bench test with its printed ratios, train_bench at batch 8 and 5
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_e
mkdir -p $OUT
timeout 1800 python bench.py --steps 20 --warmup 5 > $OUT/bench_lines.json 2> $OUT/bench.err; tail -c 1700 $OUT/bench_lines.json
python tools/engine_memory.py > $OUT/engine_memory.json 2>&1; cat $OUT/engine_memory.json
python tools/engine_memory.py --workload c4_large_pocket --batch 64 > $OUT/engine_memory_c4.json 2>&1; cat $OUT/engine_memory_c4.json
timeout 900 python -m pytest tests/test_gpu_bench_ranks.py tests/test_gpu_distributed.py tests/test_gpu_train_step.py tests/test_gpu_train_graph.py -q -s > $OUT/pytest_ranks.log 2>&1; grep -E "two ranks|passed|failed" $OUT/pytest_ranks.log
python tools/train_bench.py --batch 8 2>/dev/null | tail -1 > $OUT/train_bench_b8.json; cut -c1-400 $OUT/train_bench_b8.json | tail -c 300
python tools/train_bench.py --batch 5 2>/dev/null | tail -1 > $OUT/train_bench_b5.json; cut -c1-400 $OUT/train_bench_b5.json | tail -c 300
