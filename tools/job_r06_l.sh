# round 6, job l: A/B of the cost-weighted split against the equal-unit split on ONE box (diagnostic library, CBD_S_EQUAL_UNITS), alternating
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_l
mkdir -p $OUT
for i in 1 2 3; do for eq in 1 0; do
  CBD_S_EQUAL_UNITS=$eq python bench.py --diag-library --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps 6 --warmup 2 --pair 2 --headline-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json, os
d = json.loads(sys.stdin.read()); print('equal_units=' + os.environ['CBD_S_EQUAL_UNITS'], 'c4 bf16 pair 2:', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done; done | tee $OUT/ab.txt
for eq in 1 0; do CBD_S_EQUAL_UNITS=$eq CBD_DIAG_MIN_ROLES=4 python tools/conv_span_wg.py 2>&1 | grep -E "rep 3|by last role" | cut -c1-330 | sed "s/^/equal_units=$eq /"; done | tee $OUT/span_ab.txt
