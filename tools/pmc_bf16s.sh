# PMC passes over the C4 bf16 bench with the register-stationary kernel (one counter per pass, no tracing next to --pmc), summarised into
# profiles/<tag>_pmc_bf16s_summary.txt.      bash tools/pmc_bf16s.sh r05_b
set -x
export TAG=${1:-r05_b}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/$TAG
mkdir -p $OUT
export CBD_BF16_STATIONARY=1
for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_bf16s/$c -o p -- python bench.py --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 --steps 1 --warmup 0 --pair 0 --headline-only --no-cpu-baseline > $OUT/pmc_bf16s_$c.log 2>&1 || echo FAILED $c
done
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
tag = os.environ["TAG"]
per = defaultdict(list)
for f in glob.glob(f"gpurun_out/{tag}/pmc_bf16s/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tp_conv64s_kernel" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = ["kernel,counter,launches,sum,mean_per_launch"]
for c, v in sorted(per.items()):
    lines.append(f'"tp_conv64s_kernel",{c},{len(v)},{sum(v)},{sum(v) / len(v)}')
s = {c: sum(v) for c, v in per.items()}
if "SQ_VALU_MFMA_BUSY_CYCLES" in s and "GRBM_GUI_ACTIVE" in s:
    lines.append(f"# matrix-pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)) = {s['SQ_VALU_MFMA_BUSY_CYCLES'] / (s['GRBM_GUI_ACTIVE'] / 8 * 1024):.4f}")
if "SQ_INSTS_MFMA" in s:
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU"):
        if c in s:
            lines.append(f"# {c} per MFMA = {s[c] / s['SQ_INSTS_MFMA']:.3f}")
open(f"gpurun_out/{tag}/{tag}_pmc_bf16s_summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/pmc_bf16s
