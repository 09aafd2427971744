"""Summarise the rocprofv3 --pmc passes (one counter per pass, run separately from --kernel-trace as the pool requires) of
`bench.py --steps 1 --warmup 0` into profiles/<tag>_pmc_summary.csv and profiles/<tag>_traffic.json.
HBM bytes per tp_conv<3,3> launch = FETCH_SIZE [KB] x 2 (gfx950 reports half of wide coalesced reads, MI355X_MICROARCH.md,
HBM section) + WRITE_SIZE [KB]; matrix-pipe busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).
Usage: python tools/pmc_summary.py gpurun_out/pmc_d r01_d"""
import csv
import json
import os
import sys
from collections import defaultdict


def load(path):
    rows = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            rows[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return rows


def main():
    root, tag = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else "bench.py"
    out_rows, per = [], {}
    import glob
    for c in ("FETCH_SIZE", "WRITE_SIZE", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
        found = glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True)
        rows = load(found[0])
        for (k, cn), v in rows.items():
            if "cbd::tp_conv_kernel" not in k or "OpsF32" not in k or cn != c:
                continue
            name = "tp_conv<3,3>" if "<3, 3" in k else "tp_conv<embedding layers>"
            per.setdefault((name, c), []).extend(v)
    for (name, c), v in sorted(per.items()):
        out_rows.append((name, c, len(v), sum(v) / len(v), min(v), max(v)))
    prof = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    with open(os.path.join(prof, f"{tag}_pmc_tp_conv_summary.csv"), "w") as f:
        f.write("kernel,counter,launches,mean_per_launch,min,max\n")
        for r in out_rows:
            f.write(",".join(f'"{x}"' if isinstance(x, str) and "," in x else str(x) for x in r) + "\n")
    m = {(n, c): mean for n, c, _, mean, _, _ in out_rows}
    fetch, write = m[("tp_conv<3,3>", "FETCH_SIZE")], m[("tp_conv<3,3>", "WRITE_SIZE")]
    busy = sum(per[("tp_conv<3,3>", "SQ_VALU_MFMA_BUSY_CYCLES")]) / (sum(per[("tp_conv<3,3>", "GRBM_GUI_ACTIVE")]) / 8 * 1024)
    allf = per[("tp_conv<3,3>", "FETCH_SIZE")] + per.get(("tp_conv<embedding layers>", "FETCH_SIZE"), [])
    allw = per[("tp_conv<3,3>", "WRITE_SIZE")] + per.get(("tp_conv<embedding layers>", "WRITE_SIZE"), [])
    j = {"kernel": "tp_conv_kernel<3,3>", "hbm_bytes_per_launch": (2 * fetch + write) * 1024, "fetch_size_kb": fetch, "write_size_kb": write,
         "hbm_bytes_per_launch_all_tp_conv": (2 * sum(allf) / len(allf) + sum(allw) / len(allw)) * 1024, "launches_all_tp_conv": len(allf),
         "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE in separate passes over "
                 f"`{cmd}`; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section); mean over the "
                 f"{len(per[('tp_conv<3,3>', 'FETCH_SIZE')])} tp_conv<3,3> launches", "mfma_busy_frac": busy}
    json.dump(j, open(os.path.join(prof, f"{tag}_traffic.json"), "w"), indent=1)
    print(json.dumps(j))


if __name__ == "__main__":
    main()
