TAG=r03_l
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_op.py tests/test_gpu_finetune_loop.py "tests/test_gpu_configs.py::test_config_c5_bootstrapping_round_at_size" -q -x -m gpu 2>&1 | tail -4
timeout 600 python tools/train_profile.py --batch 8 > $OUT/train_profile_b8.txt 2>&1
head -9 $OUT/train_profile_b8.txt | cut -c1-220 | tail -3
grep -i "ReduceAdd\|indexFunc\|bincount\|hipMemcpyWithStream\|hipStreamSynchronize" $OUT/train_profile_b8.txt | cut -c1-200 | head
timeout 600 python tools/train_bench.py --batch 8 2>&1 | tail -3
timeout 600 python tools/train_bench.py --batch 32 2>&1 | tail -3
