#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) result database into a small per-kernel CSV summary.

usage: python profiles/extract_rocprof.py gpurun_out/prof_xxx/name_results.db profiles/r01_xxx_kernel_stats.csv
Durations are MICROSECONDS as stored by rocprofv3 --kernel-trace --stats (view `top_kernels`).
"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for name, calls, tot, avg, pct in rows:
            short = name if len(name) < 160 else name[:157] + "..."
            w.writerow([short, calls, f"{tot:.1f}", f"{avg:.1f}", f"{pct:.4f}"])
    print(f"wrote {out_path} ({len(rows)} kernels)")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])


def pmc_summary(db_path, out_path):
    """Per-kernel sums of the PMC counters of a `rocprofv3 --pmc X` run (view `counters_collection`).
    FETCH_SIZE / WRITE_SIZE are reported in KiB; on gfx950 FETCH_SIZE counts HALF of the bytes of wide
    coalesced reads (MI355X_MICROARCH.md, HBM section) -- the CSV keeps the raw counter value."""
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = list(cur.execute(
        "select kernel_name, counter_name, sum(value), count(*) from counters_collection "
        "group by kernel_name, counter_name order by sum(value) desc"))
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "sum_value_KiB", "dispatches", "per_dispatch_KiB"])
        for name, cname, val, n in rows:
            short = name if len(name) < 160 else name[:157] + "..."
            w.writerow([short, cname, f"{val:.1f}", n, f"{val / max(n, 1):.1f}"])
    print(f"wrote {out_path} ({len(rows)} rows)")
