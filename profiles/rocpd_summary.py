#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd` -> NAME_results.db) into the
per-kernel summary CSV committed under profiles/ (name, calls, total us, average us, percent)."""
import csv
import sqlite3
import sys


def main(db_path, out_csv):
    cur = sqlite3.connect(db_path).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "average_us", "percent"])
        for name, calls, total, avg, pct in rows:
            w.writerow([name, calls, f"{total:.3f}", f"{avg:.3f}", f"{pct:.3f}"])   # the view reports microseconds
    print(f"{out_csv}: {len(rows)} kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
