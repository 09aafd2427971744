#!/bin/bash
mkdir -p gpurun_out/r03_ae
timeout 2000 python -m pytest tests/test_gpu_train_step.py tests/test_gpu_train_op.py tests/test_gpu_finetune_loop.py "tests/test_gpu_configs.py::test_config_c5_bootstrapping_round_at_size" tests/test_gpu_distributed.py -q -x -m gpu 2>&1 | tail -3
python bench.py > gpurun_out/r03_ae/bench.json 2> gpurun_out/r03_ae/bench.err
tail -c 3000 gpurun_out/r03_ae/bench.json
