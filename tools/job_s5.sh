mkdir -p gpurun_out/s13
B="python bench.py --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps 4 --warmup 1 --headline-only --pair 2 --no-cpu-baseline"
for r in 1 2; do
CBD_BF16_STATIONARY=1 timeout 300 $B > gpurun_out/s13/bench_stat_$r.json 2> gpurun_out/s13/bench_stat.err; echo "rc $?"
tail -n 1 gpurun_out/s13/bench_stat_$r.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done
for v in 0 0 992; do CBD_BF16_DIAG=$v timeout 200 python tools/bf16s_variants.py $v 2>&1 | grep variant; done
