"""Summarise the rocprofv3 --pmc passes of tools/pmc_c4_bf16.sh (one counter per pass) for `tp_conv64_kernel` into
profiles/<tag>_pmc_bf16_c4_tp_conv64_summary.txt and profiles/<tag>_c4_bf16_traffic.json.
HBM bytes per launch = FETCH_SIZE [KB] x 2 (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md, HBM section) + WRITE_SIZE
[KB]; matrix-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).
Usage: python tools/pmc_bf16_summary.py gpurun_out/<tag>/pmc_bf16 <tag>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    root, tag = sys.argv[1], sys.argv[2]
    per = defaultdict(list)
    stationary = False
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "tp_conv64_kernel" not in k and "tp_conv64s_kernel" not in k:
                continue
            # the 74 -> 74 layers: the register-stationary kernel (tp_conv64s_kernel, the default) or the streaming kernel's <3, 3> instance
            name = "tp_conv64<3,3>" if ("<3, 3" in k or "tp_conv64s_kernel" in k) else "tp_conv64<embedding layers>"
            stationary = stationary or "tp_conv64s_kernel" in k
            per[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    lines = ["kernel,counter,launches,mean_per_launch,min,max"]
    for (name, c), v in sorted(per.items()):
        lines.append(f'"{name}",{c},{len(v)},{sum(v) / len(v)},{min(v)},{max(v)}')
    m = {k: sum(v) / len(v) for k, v in per.items()}
    K = "tp_conv64<3,3>"
    busy = sum(per[(K, "SQ_VALU_MFMA_BUSY_CYCLES")]) / (sum(per[(K, "GRBM_GUI_ACTIVE")]) / 8 * 1024)
    valu_per_mfma = m[(K, "SQ_INSTS_VALU")] / m[(K, "SQ_INSTS_MFMA")] if (K, "SQ_INSTS_MFMA") in m else None
    allf = per[(K, "FETCH_SIZE")] + per.get(("tp_conv64<embedding layers>", "FETCH_SIZE"), [])
    allw = per[(K, "WRITE_SIZE")] + per.get(("tp_conv64<embedding layers>", "WRITE_SIZE"), [])
    pair = os.environ.get("CBD_PMC_PAIR", "2")
    j = {"kernel": "tp_conv64s_kernel (register-stationary)" if stationary else "tp_conv64_kernel<3,3>", "hbm_bytes_per_launch": (2 * m[(K, "FETCH_SIZE")] + m[(K, "WRITE_SIZE")]) * 1024,
         "fetch_size_kb": m[(K, "FETCH_SIZE")], "write_size_kb": m[(K, "WRITE_SIZE")],
         "hbm_bytes_per_launch_all_tp_conv": (2 * sum(allf) / len(allf) + sum(allw) / len(allw)) * 1024, "launches_all_tp_conv": len(allf),
         "mfma_busy_frac": busy, "valu_insts_per_mfma": valu_per_mfma,
         "note": "rocprofv3 --pmc, one counter per pass, over `bench.py --workload c4_large_pocket --dtype bf16 --samples 64 --denoise-steps 40 "
                 f"--steps {pair} --warmup 0 --pair {pair} --headline-only`; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section)"}
    lines.append(f"# matrix-pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)) = {busy:.4f}; VALU per MFMA = {valu_per_mfma}")
    open(os.path.join(prof, f"{tag}_pmc_bf16_c4_tp_conv64_summary.txt"), "w").write("\n".join(lines) + "\n")
    json.dump(j, open(os.path.join(prof, f"{tag}_c4_bf16_traffic.json"), "w"), indent=1)
    print(json.dumps(j))


if __name__ == "__main__":
    main()
