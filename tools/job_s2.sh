mkdir -p gpurun_out/s2
B="python bench.py --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps 4 --warmup 1 --headline-only --pair 2 --no-cpu-baseline"
CBD_BF16_STATIONARY=0 timeout 300 $B > gpurun_out/s2/bench_stream.json 2> gpurun_out/s2/bench_stream.err; echo "rc $?"
CBD_BF16_STATIONARY=1 timeout 300 $B > gpurun_out/s2/bench_stat.json 2> gpurun_out/s2/bench_stat.err; echo "rc $?"
timeout 200 python tools/conv_clock_s.py > gpurun_out/s2/clock_s.txt 2>&1; echo "rc $?"
tail -c 1500 gpurun_out/s2/bench_stream.json; echo; tail -c 1500 gpurun_out/s2/bench_stat.json; echo; cat gpurun_out/s2/clock_s.txt
