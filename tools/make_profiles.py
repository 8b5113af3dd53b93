#!/usr/bin/env python3
"""Distil a gpurun_out/<dir> produced by the profiling recipe in DESIGN.md section 5 into profiles/r<NN>_*.
usage: make_profiles.py <gpurun_out dir> <round tag, e.g. r01>"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1].rstrip("/") + "/", sys.argv[2]
out = os.path.join(ROOT, "profiles") + "/"
KEY = "spconv_os_split_kernel<128, 128"


def mean_counter(fn, cname):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(src + fn))
            if KEY in r["Kernel_Name"] and r["Counter_Name"] == cname]
    return sum(vals) / len(vals), len(vals)


f, nf = mean_counter("fetch_counter_collection.csv", "FETCH_SIZE")
w, nw = mean_counter("write_counter_collection.csv", "WRITE_SIZE")
bench = json.loads(open(src + "bench.json").read().strip().splitlines()[-1])
pm = {"kernel": "spconv_os_split_kernel<128,128,RT=1,NW=8> (conv4 stage: 4 x K=27 residual-block layers + the K=3 tail)",
      "kernel_key": [128, 128, 27, 1], "FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w, "launches_averaged": nf,
      "correction": "FETCH_SIZE x2 (gfx950 rocprofv3 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); "
                    "WRITE_SIZE as reported; x1024 (values are KB)",
      "traffic_bytes_per_launch": int((2 * f + w) * 1024),
      "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
      "note": "mean over all <128,128> launches of a step; below the algorithmic bytes because gathered input rows are "
              "re-read out of L2 / Infinity Cache, not HBM; the packed filter bank is re-read by every workgroup out of L2",
      "command": "rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE, separate pass) --kernel-trace --output-format csv -- "
                 "python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing"}
json.dump(pm, open(out + tag + "_pmc_spconv_split.json", "w"), indent=1)
shutil.copy(src + "stats_kernel_stats.csv", out + tag + "_bench_cp_fusion_kernel_stats.csv")
shutil.copy(src + "bench.json", out + tag + "_bench_cp_fusion.json")
for fn, o in (("fetch_counter_collection.csv", "_pmc_fetch_size.csv"), ("write_counter_collection.csv", "_pmc_write_size.csv")):
    rows = list(csv.reader(open(src + fn)))
    keep = [rows[0]] + [r for r in rows[1:] if "spconv" in r[8] or "ffn_split" in r[8] or "img_proj" in r[8]]
    csv.writer(open(out + tag + o, "w", newline="")).writerows(keep)
print(json.dumps(pm, indent=1))
print("bench:", bench["value"], bench["roofline"])
