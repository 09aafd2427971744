cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_a
python bench.py > gpurun_out/r04_a/bench_lines.json 2> gpurun_out/r04_a/bench.err; tail -c 1500 gpurun_out/r04_a/bench_lines.json
bash tools/pmc_c4_bf16.sh r04_a > gpurun_out/r04_a/pmc.log 2>&1; tail -3 gpurun_out/r04_a/pmc_bf16_summary.log
