TAG=r03_i
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 2400 python -m pytest tests -q -x -m gpu --durations=5 2>&1 | tail -15 > $OUT/pytest_gpu.log
cat $OUT/pytest_gpu.log
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
d = json.load(open("$OUT/bench_default.json"))
print("headline", d["value"], d["roofline"]["frac"]); print("python_api", d["python_api"]["value"], d["python_api"]["vs_engine_level_value"], d["python_api"]["without_confidence_model"]); print("c4", d["c4_bf16"]["value"], d["c4_bf16"]["roofline"]["frac"], d["c4_bf16"]["roofline"]["algorithmic_frac"]); print(d["other_operand_modes"]); print(d["finetune"]); print(d["complex_set"]["value"], d["confidence"]["frac"])
PY
