# One GPU-box job that regenerates the round's evidence under gpurun_out/<tag>/ (copy what is to be judged into profiles/):
#   full -m gpu test log, the default bench lines (driver's flags), rocprofv3 kernel stats + TIMED-REGION-ONLY summary + PMC passes of the
#   default bench, C4 bf16 kernel stats + PMC passes, the fine-tuning step's profiles.      bash tools/final_evidence.sh r06_z
TAG=${1:-r06_z}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 1800 python -m pytest tests -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
timeout 600 python -m pytest experiments/test_role_split.py -q > $OUT/pytest_experiments.log 2>&1; tail -2 $OUT/pytest_experiments.log
python bench.py --steps 20 --warmup 5 > $OUT/bench_lines.json 2> $OUT/bench_line.err; tail -c 600 $OUT/bench_lines.json
bash tools/profile_default.sh $TAG > $OUT/profile_default.log 2>&1
bash tools/profile_c4_bf16.sh $TAG > $OUT/profile_c4.log 2>&1
bash tools/pmc_c4_bf16.sh $TAG > $OUT/pmc_c4.log 2>&1
bash tools/profile_train.sh $TAG > $OUT/profile_train.log 2>&1
bash tools/job_train_prof.sh $TAG > $OUT/job_train_prof.log 2>&1
ls -la $OUT
