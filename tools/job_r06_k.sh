# round 6, job k: cost-weighted work split of the register-stationary bf16 kernel
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_k
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_bf16.py -q 2>&1 | tail -2
CBD_DIAG_MIN_ROLES=4 python tools/conv_span_wg.py 2>&1 | grep -E "rep 3|by last role" | cut -c1-400 | tee $OUT/span4.txt
for i in 1 2; do python bench.py --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps 6 --warmup 2 --pair 2 --headline-only --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('c4 bf16 pair 2:', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"; done | tee $OUT/c4.txt
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --legs other_operand_modes 2>/dev/null | grep other_operand | cut -c1-700 | tee $OUT/c2_modes.txt
