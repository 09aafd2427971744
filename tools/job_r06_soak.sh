# round 6 soak on the final tree: fuzz campaigns vs the CPU oracle (two new seeds), the training path vs the engine, the pipelined
# sampling() soak, the bf16 repeatability soak (both bf16 kernels), f32_split and fp32 race hunts
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_soak
mkdir -p $OUT
for s in 21 22; do timeout 900 python tools/fuzz_campaign.py $s 2>&1 | tail -2 | sed "s/^/fuzz_campaign seed $s: /"; done | tee $OUT/fuzz_campaign.txt
timeout 900 python tools/fuzz_train_vs_engine.py 2>&1 | tail -2 | tee $OUT/fuzz_train_vs_engine.txt
for s in 31 32 33; do timeout 600 python tools/fuzz_pipeline.py $s 6 2>&1 | tail -1 | sed "s/^/fuzz_pipeline seed $s: /"; done | tee $OUT/fuzz_pipeline.txt
timeout 900 python tools/bf16_repeat.py 6 bf16 2>&1 | tail -2 | tee $OUT/bf16_repeat_stationary.txt
CBD_BF16_STATIONARY=0 timeout 900 python tools/bf16_repeat.py 4 bf16 2>&1 | tail -2 | tee $OUT/bf16_repeat_streaming.txt
timeout 600 python tools/race_hunt.py 6 f32_split 2>&1 | tail -2 | tee $OUT/race_hunt_f32_split.txt
timeout 600 python tools/race_hunt.py 6 f32 2>&1 | tail -2 | tee $OUT/race_hunt_f32.txt
