# round 6, job x: evidence of the C4 bf16 leg at its new configuration (eight complexes per launch) + the leg itself
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=r06_x
OUT=gpurun_out/$TAG
mkdir -p $OUT
bash tools/profile_c4_bf16.sh $TAG 8 > $OUT/profile_c4.log 2>&1
bash tools/pmc_c4_bf16.sh $TAG 8 > $OUT/pmc_c4.log 2>&1; tail -1 $OUT/pmc_bf16_summary.log | cut -c1-700
cp profiles/${TAG}_c4_bf16_traffic.json profiles/${TAG}_pmc_bf16_c4_tp_conv64_summary.txt $OUT/ 2>/dev/null
for i in 1 2; do python bench.py --steps 4 --warmup 1 --no-cpu-baseline --legs c4_bf16 2>/dev/null | grep '"leg": "c4_bf16"' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('c4 leg:', d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_source'], d['config']['co_scheduled_complexes'])"; done | tee $OUT/c4_leg.txt
