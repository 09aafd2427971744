# round 6, job n: unit-cost weights of the bf16 kernel's work split, C4 leg, alternating settings on one box
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_n
mkdir -p $OUT
for i in 1 2; do for w in "70,64,68,67" "71,64,69,68" "72,64,70,68" "74,64,70,69" "70,64,70,68" "76,64,72,70" "70,64,68,66"; do
  export CBD_S_WEIGHTS=$w; python bench.py --diag-library --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps 6 --warmup 2 --pair 2 --headline-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json, os
d = json.loads(sys.stdin.read()); print('weights', os.environ['CBD_S_WEIGHTS'], 'c4 bf16 pair 2:', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done; done | tee $OUT/weights.txt
