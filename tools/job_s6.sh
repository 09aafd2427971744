mkdir -p gpurun_out/s14
timeout 150 python tools/bf16_stat_check.py --workload c2_dockgen_median --batch 4 > gpurun_out/s14/check.json 2> gpurun_out/s14/check.err; echo "rc $?"; cat gpurun_out/s14/check.json
B4="python bench.py --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps 4 --warmup 1 --headline-only --pair 2 --no-cpu-baseline"
B2="python bench.py --dtype bf16 --steps 8 --warmup 2 --headline-only --no-cpu-baseline"
for st in 1 0; do
CBD_BF16_STATIONARY=$st timeout 300 $B4 2> gpurun_out/s14/err.txt | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C4 stationary=$st', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
CBD_BF16_STATIONARY=$st timeout 300 $B2 2> gpurun_out/s14/err.txt | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C2 stationary=$st', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done
