# Diagnostic PMC passes over the bf16 C4 bench (one short run per counter group; SQ counters only, no tracing).
set -x
export TAG=${1:-r02c}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
CMD="python bench.py --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps 1 --warmup 0 --pair 0 --headline-only --diag-library"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/$TAG/pmc_bf16/g$i -o p -- $CMD > gpurun_out/$TAG/pmc_bf16_g$i.log 2>&1 || echo FAILED $grp
done
python - > gpurun_out/$TAG/pmc_bf16_summary.txt <<'PY'
import csv, glob, os, collections
root='gpurun_out/%s/pmc_bf16' % os.environ.get('TAG', 'r02c')
res={}
for f in glob.glob(os.path.join(root,'**','*counter_collection.csv'),recursive=True):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'tp_conv64_kernel<3, 3' in k:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for c,v in acc.items():
        res[c]=(len(v), sum(v)/len(v))
for c,v in sorted(res.items()): print(c, v)
PY
cat gpurun_out/$TAG/pmc_bf16_summary.txt
rm -rf gpurun_out/$TAG/pmc_bf16
