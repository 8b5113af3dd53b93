#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (…_results.db) into the per-kernel summary CSV we commit under
profiles/ (same columns as `rocprofv3 --stats`: name, calls, total us, average us, percentage)."""
import csv
import sqlite3
import sys


def main(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    # durations are stored in ns in some builds and us in others: normalise on the total
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDuration", "AverageDuration", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], "%.3f" % r[2], "%.3f" % r[3], "%.4f" % r[4]])
    print("wrote %s (%d kernels)" % (out, len(rows)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
