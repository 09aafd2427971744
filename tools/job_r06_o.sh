# round 6, job o: wave 1's last LDS tile moved to wave 3 (product library) against the old assignment (diagnostic library built with
# -DCBD_S_W3_MID27=0), alternating on one box; correctness of the new assignment first
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_o
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_bf16.py -q 2>&1 | tail -2
for i in 1 2 3; do for lib in "" "--diag-library"; do
  python bench.py $lib --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps 6 --warmup 2 --pair 2 --headline-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('lib=' + ('old (MID27=0)' if '$lib' else 'new (MID27=1)'), 'c4 bf16 pair 2:', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
done; done | tee $OUT/ab.txt
