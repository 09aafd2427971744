# PMC passes over the C4-as-specified bf16 bench (BASELINE.json configs[3]; one counter per pass, no tracing next to --pmc), summarised
# into profiles/<tag>_pmc_bf16_c4_tp_conv64_summary.txt and profiles/<tag>_c4_bf16_traffic.json (what bench.py's c4 leg reports as
# roofline.traffic).      bash tools/pmc_c4_bf16.sh r04_a
set -x
export TAG=${1:-r04_a}
export CBD_PMC_PAIR=${2:-8}      # complexes per launch: the c4_bf16 leg of bench.py co-schedules eight since round 6 (two in round 5)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_bf16/$c -o p -- python bench.py --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps $CBD_PMC_PAIR --warmup 0 --pair $CBD_PMC_PAIR --headline-only --no-cpu-baseline > $OUT/pmc_bf16_$c.log 2>&1 || echo FAILED $c
done
python tools/pmc_bf16_summary.py $OUT/pmc_bf16 $TAG > $OUT/pmc_bf16_summary.log 2>&1
cat $OUT/pmc_bf16_summary.log
cp profiles/${TAG}_pmc_bf16_c4_tp_conv64_summary.txt profiles/${TAG}_c4_bf16_traffic.json $OUT/ 2>/dev/null
rm -rf $OUT/pmc_bf16
