#!/usr/bin/env python3
"""Average duration per (kernel, grid) of a rocprofv3 --kernel-trace --output-format csv run.
usage: trace_avg.py <dir or kernel_trace.csv> [substring]"""
import collections
import csv
import glob
import os
import sys

path = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "df3d"
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[-1]
agg = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    if sub not in r["Kernel_Name"]:
        continue
    key = (r["Kernel_Name"][:70], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""))
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    v = v[len(v) // 4:]                      # drop warm-up launches
    print("%-70s grid %8s x%3s n %4d avg %8.1f us" % (k[0], k[1], k[2], len(v), sum(v) / len(v)))
