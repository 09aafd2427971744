TAG=r03_e
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
for D in 0 1 24; do
  CBD_BF16_DIAG=$D timeout 600 python bench.py --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps 4 --warmup 1 --headline-only > $OUT/c4_diag$D.json 2> $OUT/c4_diag$D.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/c4_diag$D.json"))
    print("diag $D:", d["value"], "poses/s  algorithmic frac", d["roofline"]["algorithmic_frac"], "avg_launch_ms", d["roofline"]["avg_launch_ms"])
except Exception as e:
    print("diag $D: failed", e)
PY
done
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_distributed.py "tests/test_gpu_configs.py::test_config_c5_bootstrapping_round_at_size" -q --durations=5 -m gpu 2>&1 | tail -25 > $OUT/pytest_new.log
cat $OUT/pytest_new.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
print("headline", d["value"], d["roofline"]["frac"]); print("python_api", d["python_api"]); print("c4", d["c4_bf16"]["value"], d["c4_bf16"]["roofline"]["frac"]); print(d["other_operand_modes"])
PY
