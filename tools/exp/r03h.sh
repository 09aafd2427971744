TAG=r03_h
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
for D in 0 8 24; do
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
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_train_step.py tests/test_gpu_train_op.py tests/test_gpu_finetune_loop.py -q -x -m gpu 2>&1 | tail -5
timeout 600 python tools/train_profile.py --batch 8 > $OUT/train_profile_b8.txt 2>&1
head -12 $OUT/train_profile_b8.txt | cut -c1-220
grep -i "ReduceAdd\|indexFunc" $OUT/train_profile_b8.txt | head
