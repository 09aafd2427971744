timeout 200 python tools/bf16_stat_check.py --workload c2_dockgen_median --batch 4 2>/dev/null | cut -c1-260
bash tools/job_prof_bf16s.sh 1 2>&1 | grep -E "64s_kernel<0>:"
CBD_BF16_DIAG=6 CBD_DIAG_MIN_ROLES=4 timeout 300 python tools/conv_clock_s.py 2>&1 | grep "^wave"
