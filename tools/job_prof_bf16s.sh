# kernel statistics of the C4 bf16 run with the streaming (0) and the register-stationary (1) kernel:   gpurun -- bash tools/job_prof_bf16s.sh [modes]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
MODES=${1:-"0 1"}
for r in $MODES; do
  OUT=gpurun_out/prof_bf16s_$r
  rm -rf $OUT; mkdir -p $OUT
  export CBD_BF16_STATIONARY=$r
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python3 bench.py --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps 2 --warmup 1 --headline-only --pair 2 --no-cpu-baseline > $OUT/log.txt 2>&1
  find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
  find $OUT/prof -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
  python3 - "$OUT/kernel_trace.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "tp_conv64" in n:
        d[n.split("(")[0][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in d.items():
    v.sort()
    big = [x for x in v if x > 0.5 * v[-1]]
    if "<3, 3" in n or "64s" in n:
        raw = [(int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if r["Kernel_Name"].startswith(n[:40]) and ("<3, 3" in r["Kernel_Name"] or "64s" in r["Kernel_Name"])]
        raw.sort()
        open(sys.argv[1].replace("kernel_trace.csv", "conv33_durations_us.txt"), "w").write("\n".join(f"{x[1]:.1f}" for x in raw))
    print(f"{n}: {len(v)} launches, total {sum(v)/1e3:.1f} ms, max {v[-1]:.0f} us, mean of the large launches ({len(big)}) {sum(big)/len(big):.0f} us")
PY
  rm -rf $OUT/prof $OUT/kernel_trace.csv
  echo "== stationary $r"; head -6 $OUT/kernel_stats.csv | cut -c1-160; tail -1 $OUT/log.txt | cut -c1-120
done
