mkdir -p gpurun_out/r5a
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py tests/test_gpu_confidence.py tests/test_gpu_edge_cases.py tests/test_gpu_train_step.py -x -q > gpurun_out/r5a/pytest.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r5a/pytest.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r5a/bench_lines.json 2> gpurun_out/r5a/bench.err; echo "bench rc $?"
python - <<'PY'
import json
for l in open('gpurun_out/r5a/bench_lines.json'):
    l=l.strip()
    if not l.startswith('{'): continue
    d=json.loads(l)
    if 'leg' in d:
        print('leg', d['leg'], {k: d[k] for k in d if k in ('value','poses_per_s','ms_per_40_poses','frac','ms_per_step','vs_engine_level_value')} , (d.get('roofline') or {}).get('frac'))
    else:
        print('HEAD', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d.get('legs'))
PY
