# instruction-cache counters of the C4 bf16 run (register-stationary kernel; four wave programs of 5-12 KB each per workgroup) -- one group per pass
set -x
TAG=${1:-r06_m}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
CMD="python3 bench.py --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps 1 --warmup 0 --pair 2 --headline-only --no-cpu-baseline"
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc/g$i -o p -- $CMD > $OUT/pmc_g$i.log 2>&1 || echo FAILED $grp
done
python3 - > $OUT/icache_summary.txt <<'PY'
import csv, glob, os, collections
root = 'gpurun_out/%s/pmc' % os.environ.get('TAG', 'r06_m')
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        name = 'tp_conv64s' if 'tp_conv64s_kernel' in k else 'tp_conv64<emb>' if 'tp_conv64_kernel' in k else None
        if name:
            res[name][r['Counter_Name']].append(float(r['Counter_Value']))
for name, d in res.items():
    print(name)
    for c, v in sorted(d.items()):
        print(f'  {c:32s} launches {len(v):5d}  mean per launch {sum(v) / len(v):16.1f}')
PY
cat $OUT/icache_summary.txt
rm -rf $OUT/pmc
