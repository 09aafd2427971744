mkdir -p gpurun_out/s12
timeout 150 python tools/bf16_stat_check.py --workload c2_dockgen_median --batch 4 > gpurun_out/s12/check.json 2> gpurun_out/s12/check.err; echo "rc $?"; cat gpurun_out/s12/check.json
for v in 0 992 16; do CBD_BF16_DIAG=$v timeout 200 python tools/bf16s_variants.py $v 2>&1 | grep variant; done
