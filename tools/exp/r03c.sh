TAG=r03_c
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
CBD_BF16_DIAG=4 timeout 300 python tools/conv_clock.py > $OUT/conv_clock_bf16.txt 2>&1
tail -3 $OUT/conv_clock_bf16.txt
timeout 1500 python -m pytest tests/test_gpu_configs.py -x -q --durations=5 2>&1 | tail -25 > $OUT/pytest_configs.log
cat $OUT/pytest_configs.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
tail -c 6000 $OUT/bench_default.json
tail -5 $OUT/bench_default.err
