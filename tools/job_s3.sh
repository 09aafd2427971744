mkdir -p gpurun_out/s11
B="python bench.py --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps 4 --warmup 1 --headline-only --pair 2 --no-cpu-baseline"
timeout 150 python tools/bf16_stat_check.py --workload c2_dockgen_median --batch 4 > gpurun_out/s11/check.json 2> gpurun_out/s11/check.err; echo "rc $?"; cat gpurun_out/s11/check.json
CBD_BF16_STATIONARY=1 timeout 300 $B > gpurun_out/s11/bench_stat.json 2> gpurun_out/s11/bench_stat.err; echo "rc $?"
timeout 200 python tools/conv_clock_s.py > gpurun_out/s11/clock_s.txt 2>&1; echo "rc $?"
tail -n 1 gpurun_out/s11/bench_stat.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"; cat gpurun_out/s11/clock_s.txt
