# non-cbd GPU kernels of the fine-tuning step (launches per step and share of kernel time):  gpurun -- bash tools/job_train_aten.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/train_aten; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o p -- python3 tools/train_profile.py --batch 8 --plain > $OUT/log.txt 2>&1
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv; rm -rf $OUT/p
python3 - $OUT/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
steps = max(int(r["Calls"]) for r in rows if "tp_train_dw_kernel<3, 3>" in r["Name"]) / 5
non = [r for r in rows if "cbd::" not in r["Name"]]
print(f"steps {steps:.0f}; kernel time per step {tot/steps/1e6:.2f} ms; launches per step {sum(int(r['Calls']) for r in rows)/steps:.0f}; "
      f"non-cbd: {sum(int(r['Calls']) for r in non)/steps:.0f} launches, {sum(float(r['TotalDurationNs']) for r in non)/tot*100:.2f} % of kernel time")
for r in non[:28]:
    print(f"  {int(r['Calls'])/steps:6.1f}/step {float(r['TotalDurationNs'])/tot*100:5.2f}%  {r['Name'][:140]}")
PY
python3 tools/train_profile.py --batch 8 --regions 2>&1 | grep -A45 "GPU launches by"
