# round 6, job d: training tests after the stream changes; fine-tuning leg with the torsion head on the side stream; dW chunk sweep
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_d
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_graph.py tests/test_gpu_train_op.py tests/test_gpu_train_distributed.py tests/test_gpu_finetune_loop.py -q > $OUT/pytest_train.log 2>&1; tail -4 $OUT/pytest_train.log
for mode in "0 96" "1 96" "1 48" "1 64" "1 160"; do
  set -- $mode
  echo "== CBD_TRAIN_TWO_STREAMS=$1 CBD_DW_MAX_CHUNKS=$2"
  CBD_TRAIN_TWO_STREAMS=$1 CBD_DW_MAX_CHUNKS=$2 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --legs finetune,finetune_b5 2> $OUT/bench_$1_$2.err | grep '"leg": "finetune' | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln)
    print(d['leg'], d['ms_per_step'], d['ms_per_step_blocks'], d['roofline']['frac'], {k: v['ms_per_step'] for k, v in d['roofline']['kernels'].items()})
"
done | tee $OUT/finetune_modes.txt
