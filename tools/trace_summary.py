#!/usr/bin/env python3
"""Per-step summary of a rocprofv3 --kernel-trace CSV of bench.py: kernel-busy time, launch count and the top
kernels of the steady-state steps (steps are delimited by the voxelizer's first kernel).
usage: trace_summary.py <kernel_trace.csv> [first_step last_step] [--csv out.csv]"""
import collections
import csv
import sys


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    out_csv = sys.argv[sys.argv.index("--csv") + 1] if "--csv" in sys.argv else None
    if out_csv in args:
        args.remove(out_csv)
    rows = list(csv.DictReader(open(args[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    first = [i for i, r in enumerate(rows) if "vox_insert" in r["Kernel_Name"]]
    lo = int(args[1]) if len(args) > 1 else max(len(first) - 16, 0)
    hi = int(args[2]) if len(args) > 2 else len(first) - 2
    tot, cnt = collections.Counter(), collections.Counter()
    busy = wall = nk = 0
    for s in range(lo, hi + 1):
        seg = rows[first[s]:first[s + 1]]
        wall += int(rows[first[s + 1]]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"])
        for r in seg:
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            busy += d
            tot[r["Kernel_Name"]] += d
            cnt[r["Kernel_Name"]] += 1
        nk += len(seg)
    ns = hi - lo + 1
    print("steps %d..%d: per step wall(profiled) %.1f us, kernel busy %.1f us, %d kernel launches" % (
        lo, hi, wall / ns / 1e3, busy / ns / 1e3, nk // ns))
    lines = [("Name", "CallsPerStep", "UsPerStep", "AvgUs", "Percent")]
    for k, v in tot.most_common():
        lines.append((k, "%.2f" % (cnt[k] / ns), "%.2f" % (v / ns / 1e3), "%.2f" % (v / cnt[k] / 1e3),
                      "%.2f" % (100.0 * v / busy)))
    for l in lines[1:41]:
        print("%-96s %6s calls %9s us/step (avg %8s us) %6s%%" % (l[0][:96], l[1], l[2], l[3], l[4]))
    if out_csv:
        with open(out_csv, "w", newline="") as f:
            csv.writer(f).writerows(lines)
        print("wrote", out_csv)


if __name__ == "__main__":
    main()
