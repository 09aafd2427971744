"""Kernel summary of the TIMED REGION of a bench.py run from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d DIR -o p -- python3 bench.py --steps 20 --warmup 5 --headline-only --mark-timed-region > line.json
    python tools/timed_region_stats.py DIR line.json profiles/r06_x        ->  profiles/r06_x_timed_kernel_stats.csv, _timed_recompute.json

bench.py --mark-timed-region dispatches a marker kernel (an in-place add on int16: nothing else in the process launches it) right before
and right after the timed region.  This tool keeps the dispatches that START after the first marker ended and END before the second one
started, writes their per-kernel totals (same columns as rocprofv3 --stats) and RECOMPUTES the bench line's roofline from them:
    frac = executed TFLOP of the line / sum of tp_conv* durations of this CSV / peak
which must agree with `roofline.frac` of the line printed under the same run (HIP events) -- the file says by how much."""
import csv
import glob
import json
import os
import sys


def main():
    trace_dir, line_file, out_prefix = sys.argv[1], sys.argv[2], sys.argv[3]
    files = glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True)
    assert files, "no kernel trace under " + trace_dir
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    marks = [(a, b) for a, b, n in rows if "short" in n and ("add" in n.lower() or "Add" in n)]
    assert len(marks) == 2, f"expected two marker dispatches, found {len(marks)}: was bench.py run with --mark-timed-region?"
    lo, hi = marks[0][1], marks[1][0]
    agg = {}
    for a, b, n in rows:
        if a >= lo and b <= hi:
            t, c, mn, mx = agg.get(n, (0, 0, 1 << 62, 0))
            agg[n] = (t + (b - a), c + 1, min(mn, b - a), max(mx, b - a))
    total = sum(t for t, _, _, _ in agg.values())
    with open(out_prefix + "_timed_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, (t, c, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            w.writerow([n, c, t, round(t / c, 1), round(100.0 * t / total, 4), mn, mx])
    line = None
    for ln in open(line_file):
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            line = json.loads(ln)
    detail = None
    for ln in open(line_file):
        ln = ln.strip()
        if ln.startswith("{") and '"headline_detail"' in ln:
            detail = json.loads(ln)
    rf = (detail or line)["roofline"]
    conv_ns = sum(t for n, (t, _, _, _) in agg.items() if "tp_conv" in n)
    conv_calls = sum(c for n, (_, c, _, _) in agg.items() if "tp_conv" in n)
    tflop = rf.get("executed_tflop_total", rf["executed_gflop_per_launch"] * rf["launches"] / 1e3)
    frac = tflop / (conv_ns * 1e-9) / rf["peak"]
    rec = {"what": "roofline.frac of the bench line recomputed from the rocprofv3 kernel trace of the SAME run, timed region only "
                   "(dispatches between the two marker kernels of bench.py --mark-timed-region)",
           "bench_line": {k: line[k] for k in ("value", "ms_per_step", "steps", "warmup")}, "bench_roofline_frac_hip_events": rf["frac"],
           "executed_tflop_total": tflop, "tp_conv_launches_in_trace": conv_calls, "tp_conv_launches_in_line": rf["launches"],
           "tp_conv_ms_total_trace": round(conv_ns * 1e-6, 3), "tp_conv_ms_total_hip_events": rf.get("tp_conv_ms_total"),
           "peak_tflops": rf["peak"], "frac_recomputed_from_trace": round(frac, 4), "relative_difference": round(frac / rf["frac"] - 1.0, 5),
           "timed_region_ms_trace": round((hi - lo) * 1e-6, 3), "gpu_busy_share_of_region": round(total / (hi - lo), 4),
           "kernels_in_region": len(agg), "dispatches_in_region": sum(c for _, c, _, _ in agg.values())}
    json.dump(rec, open(out_prefix + "_timed_recompute.json", "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
