#!/usr/bin/env python3
"""Average PMC counter values per kernel from rocprofv3 --pmc ... --output-format csv runs.
usage: pmc_avg.py <dir> [<dir> ...] [--match substring]"""
import collections
import csv
import glob
import os
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
match = sys.argv[sys.argv.index("--match") + 1] if "--match" in sys.argv else "os_split"
if match in args:
    args.remove(match)
tab = collections.OrderedDict()
for d in args:
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            if match not in r["Kernel_Name"]:
                continue
            key = r["Kernel_Name"][:60] + " grid " + r["Grid_Size"]
            tab.setdefault(key, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, c in tab.items():
    print(k)
    for name, v in c.items():
        print("    %-28s %16.0f  (n=%d)" % (name, sum(v) / len(v), len(v)))
