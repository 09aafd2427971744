# round 6, job f: the complete -m gpu suite on the final tree (after the hub.grads stream wait) + smoke
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_f
mkdir -p $OUT
timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_train_step.py -q -k "repeatable or reference or look_ahead" 2>&1 | tail -1; done
CBD_TRAIN_TWO_STREAMS=1 timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --legs finetune,finetune_b5 2>/dev/null | grep '"leg": "finetune' | cut -c1-260
