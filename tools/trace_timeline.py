#!/usr/bin/env python3
"""One steady-state step of a rocprofv3 --kernel-trace CSV as a timeline: start (us from the step's first kernel), duration,
queue, kernel -- to see what overlaps what and where the GPU idles.  usage: trace_timeline.py <kernel_trace.csv> [step]"""
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    first = [i for i, r in enumerate(rows) if "vox_insert" in r["Kernel_Name"]]
    s = int(sys.argv[2]) if len(sys.argv) > 2 else len(first) - 3
    seg = rows[first[s]:first[s + 1]]
    t0 = int(seg[0]["Start_Timestamp"])
    queues = {}
    end_prev = t0
    print("step %d: %d kernels, %.1f us from first start to next step's first start" % (
        s, len(seg), (int(rows[first[s + 1]]["Start_Timestamp"]) - t0) / 1e3))
    for r in seg:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        q = queues.setdefault(r.get("Queue_Id", "?"), len(queues))
        gap = (st - end_prev) / 1e3
        end_prev = max(end_prev, en)
        name = r["Kernel_Name"].replace("void ", "").replace("df3d::", "")
        print("%9.1f %8.1f q%d %s%s" % ((st - t0) / 1e3, (en - st) / 1e3, q, ("[idle %.1f] " % gap) if gap > 2.0 else "", name[:110]))


if __name__ == "__main__":
    main()
