# round 6, job g: the ADVICE ordering scenario as a test + its sensitivity check (ordering switched off = the pre-round-6 state)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06_g
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_finetune_loop.py tests/test_gpu_complex_set.py -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 900 python tools/check_upload_ordering.py 6 > $OUT/ordering_sensitivity.txt 2>&1; tail -3 $OUT/ordering_sensitivity.txt
