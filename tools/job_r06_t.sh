# round 6, job t: per-XCD rates in the work split (diagnostic library built with a rate table) against uniform rates (product), alternating
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_t
mkdir -p $OUT
for i in 1 2 3; do for lib in "" "--diag-library"; do
  python bench.py $lib --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps 8 --warmup 2 --pair 8 --headline-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('rates ' + ('TABLE' if '$lib' else 'uniform'), 'c4 bf16 pair 8:', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done; done | tee $OUT/ab.txt
