# round 6, job q: XCD-contiguous pieces (diagnostic library built with -DCBD_S_XCD_MAP=1) against piece = blockIdx (product), alternating
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_q
mkdir -p $OUT
for i in 1 2 3; do for lib in "" "--diag-library"; do
  python bench.py $lib --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps 6 --warmup 2 --pair 2 --headline-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('pieces ' + ('XCD-contiguous' if '$lib' else 'round-robin over XCDs'), 'c4 bf16 pair 2:', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done; done | tee $OUT/ab.txt
