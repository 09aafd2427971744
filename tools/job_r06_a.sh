# round 6, job a: the -m gpu suite + the role-split experiment on the diagnostic library after the product/diagnostic split
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_a
mkdir -p $OUT
timeout 1500 python -m pytest tests -q -m gpu -x > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
timeout 600 python -m pytest experiments/test_role_split.py -q > $OUT/pytest_experiments.log 2>&1; tail -3 $OUT/pytest_experiments.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -4 $OUT/smoke.log
