timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
for pr in 0 2 4; do
  echo "== pair $pr c4 (default = stationary)"
  timeout 500 python bench.py --headline-only --no-cpu-baseline --pair $pr --workload c4_large_pocket --samples 64 --denoise-steps 40 --dtype bf16 --steps 6 --warmup 2 2>/dev/null | python -c "import sys,json; [print({k:r[k] for k in ('value','ms_per_step')}, r['roofline']['frac']) for r in (json.loads(l) for l in sys.stdin if l.startswith('{')) if 'value' in r]"
done
