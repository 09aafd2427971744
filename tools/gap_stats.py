"""GPU busy time and idle gaps from a rocprofv3 --kernel-trace CSV (one process, one GPU): is a launch-heavy loop bound by the kernels,
by the gaps between dependent kernels, or by the host?      python tools/gap_stats.py <kernel_trace.csv> [--tail 0.5]"""
import argparse
import csv
import collections


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--tail", type=float, default=0.5, help="analyse the last fraction of the trace (skips warm-up)")
    ap.add_argument("--after-largest-gap", action="store_true", help="analyse from the first kernel behind the largest idle gap of the tail "
                    "(a timed run that follows a host-side build) and list its largest gaps with their position in the run")
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    t0, t1 = rows[0][0], rows[-1][1]
    cut = t1 - (t1 - t0) * a.tail
    rows = [r for r in rows if r[0] >= cut]
    if a.after_largest_gap:
        end, best, at = rows[0][1], 0, 0
        for i, (s0, e0, _) in enumerate(rows):
            if s0 - end > best:
                best, at = s0 - end, i
            end = max(end, e0)
        rows = rows[at:]
        print(f"region: from kernel {at} behind a gap of {best / 1e6:.1f} ms")
    span = rows[-1][1] - rows[0][0]
    busy, end, gaps = 0, rows[0][0], []
    for s, e, name in rows:
        if s > end:
            gaps.append((s - end, name))
        busy += max(0, e - max(s, end))
        end = max(end, e)
    print(f"kernels {len(rows)}, span {span / 1e6:.1f} ms, busy {busy / 1e6:.1f} ms ({busy / span:.3f}), idle {(span - busy) / 1e6:.1f} ms")
    g = sorted(x for x, _ in gaps)
    if g:
        print(f"gaps: n {len(g)}, median {g[len(g) // 2] / 1e3:.1f} us, p90 {g[int(len(g) * 0.9)] / 1e3:.1f} us, max {g[-1] / 1e3:.1f} us")
        for lo, hi in ((0, 5), (5, 10), (10, 20), (20, 50), (50, 200), (200, 1e9)):
            sel = [x for x in g if lo * 1e3 <= x < hi * 1e3]
            print(f"  {lo:>4}-{hi if hi < 1e9 else 'inf':>4} us: n {len(sel):6d}  total {sum(sel) / 1e6:8.2f} ms")
        by = collections.Counter()
        for x, name in gaps:
            if x >= 20e3:
                by[name[:70]] += x
        print("idle time before kernels (gaps >= 20 us), top 12:")
        for name, x in by.most_common(12):
            print(f"  {x / 1e6:8.2f} ms  {name}")
    if a.after_largest_gap:
        print("largest gaps of the region (offset into the region, gap, kernel that ends it):")
        end = rows[0][0]
        big = []
        for s0, e0, name in rows:
            if s0 > end:
                big.append((s0 - end, end - rows[0][0], name))
            end = max(end, e0)
        for x, off, name in sorted(big, reverse=True)[:25]:
            print(f"  +{off / 1e6:9.2f} ms  {x / 1e3:9.1f} us  {name[:80]}")
    dur = collections.Counter()
    cnt = collections.Counter()
    for s, e, name in rows:
        dur[name[:70]] += e - s
        cnt[name[:70]] += 1
    print("kernel time, top 15:")
    for name, x in dur.most_common(15):
        print(f"  {x / 1e6:8.2f} ms  n {cnt[name]:6d}  {name}")


if __name__ == "__main__":
    main()
