TAG=r03_g
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
for D in 0 8; do
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
CBD_BF16_DIAG=4 timeout 300 python tools/conv_clock.py 2>&1 | tail -3
timeout 2400 python -m pytest tests -q -x -m gpu --durations=8 2>&1 | tail -22 > $OUT/pytest_gpu.log
cat $OUT/pytest_gpu.log
