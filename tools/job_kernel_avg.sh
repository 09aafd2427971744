# average duration of the training kernels under rocprofv3:  gpurun -- bash tools/job_kernel_avg.sh [batch]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
B=${1:-32}
OUT=gpurun_out/kernel_avg; rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p -o p -- python3 tools/train_profile.py --batch $B --plain > $OUT/log.txt 2>&1
f=$(find $OUT/p -name "*kernel_stats.csv" | head -1)
python3 - "$f" $B <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "tp_train_" in r["Name"] and "<3, 3>" in r["Name"] or "partial_reduce" in r["Name"]:
        print("batch", sys.argv[2], r["Name"][10:42], "avg us", round(float(r["AverageNs"]) / 1e3, 1))
PY
grep "^batch" $OUT/log.txt
rm -rf $OUT/p
