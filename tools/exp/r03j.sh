TAG=r03_k
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
export CBD_BF16_KERNEL=1
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -x -m gpu 2>&1 | tail -12
for K in 1 0; do
  CBD_BF16_KERNEL=$K timeout 600 python bench.py --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps 4 --warmup 1 --headline-only > $OUT/c4_k$K.json 2> $OUT/c4_k$K.err
  tail -3 $OUT/c4_k$K.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/c4_k$K.json"))
    print("kernel $K:", d["value"], "poses/s  algorithmic frac", d["roofline"]["algorithmic_frac"], "avg_launch_ms", d["roofline"]["avg_launch_ms"])
except Exception as e:
    print("kernel $K: failed", e)
PY
done
CBD_BF16_KERNEL=1 CBD_BF16_DIAG=4 timeout 300 python tools/conv_clock.py 2>&1 | tail -3
