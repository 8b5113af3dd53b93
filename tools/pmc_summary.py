#!/usr/bin/env python3
"""Per-kernel summary of the rocprofv3 passes written by tools/profile_r04.sh: average duration (kernel trace), HBM bytes
(FETCH_SIZE x calibration factor, WRITE_SIZE; both in KB per the guide) and SQ counters, grouped by (kernel name, grid).
usage: pmc_summary.py <gpurun_out/prof_r04x> [name filter regex] > profiles/r04_pmc_by_kernel.json"""
import collections
import csv
import json
import os
import re
import sys

src = sys.argv[1].rstrip("/") + "/"
flt = re.compile(sys.argv[2]) if len(sys.argv) > 2 else re.compile("df3d")


def rows(fn):
    p = src + fn
    return list(csv.DictReader(open(p))) if os.path.exists(p) else []


def short(name):
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void ", "").replace("df3d::", "")


dur = collections.defaultdict(list)
for r in rows("trace/stats_kernel_trace.csv"):
    if flt.search(r["Kernel_Name"]):
        if "Grid_Size" in r:
            g, w = int(r["Grid_Size"]), int(r["Workgroup_Size"])
        else:
            g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            w = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
        dur[(short(r["Kernel_Name"]), g // max(w, 1))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for d, f in (("fetch", "fetch"), ("write", "write"), ("sq", "sq"), ("sq2", "sq2")):
    for r in rows("%s/%s_counter_collection.csv" % (d, f)):
        if flt.search(r["Kernel_Name"]):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
            cnt[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for c in ("VGPR_Count", "Accum_VGPR_Count", "LDS_Block_Size"):
                if c in r and r[c] not in ("", None):
                    cnt[key]["_" + c] = [float(r[c])]
# calibration factors of the fetch counter (tools/ubench/pmc_calib.py: 2^21 rows of 512 B gathered once each)
N_CAL = 1 << 21
cal = {}
for d, f, cname, expect in (("calib_fetch", "fetch", "FETCH_SIZE", (N_CAL * 512 + N_CAL * 4) / 1024.0),
                            ("calib_write", "write", "WRITE_SIZE", 2 * N_CAL * 512 / 1024.0)):
    vals = [float(r["Counter_Value"]) for r in rows("%s/%s_counter_collection.csv" % (d, f))
            if r["Counter_Name"] == cname and "spconv_os" in r["Kernel_Name"]
            and int(r["Grid_Size"]) // int(r["Workgroup_Size"]) == N_CAL // 128]
    if vals:
        cal[cname] = expect / (sum(vals) / len(vals))
out = {"calibration_factor": {k: round(v, 4) for k, v in cal.items()}, "kernels": []}
for key in sorted(dur, key=lambda k: -sum(dur[k])):
    name, wgs = key
    e = {"kernel": name, "workgroups": wgs, "launches": len(dur[key]), "avg_us": round(sum(dur[key]) / len(dur[key]), 2),
         "total_us": round(sum(dur[key]), 1)}
    c = cnt.get(key, {})
    mean = lambda k: (sum(c[k]) / len(c[k])) if c.get(k) else None          # noqa: E731
    f, w = mean("FETCH_SIZE"), mean("WRITE_SIZE")
    if f is not None:
        e["hbm_fetch_bytes"] = int(f * 1024 * cal.get("FETCH_SIZE", 1.0))
    if w is not None:
        e["hbm_write_bytes"] = int(w * 1024 * cal.get("WRITE_SIZE", 1.0))
    if f is not None and w is not None:
        e["hbm_frac_of_8TBs"] = round((e["hbm_fetch_bytes"] + e["hbm_write_bytes"]) / (e["avg_us"] * 1e-6) / 8e12, 4)
    wc, busy = mean("SQ_WAVE_CYCLES"), mean("SQ_BUSY_CYCLES")
    if wc and busy:
        e["waves_per_simd"] = round(wc / busy / 4.0, 2) if busy else None       # SQ_BUSY_CYCLES counts per SE ... kept as a ratio
        e["wait_any_frac"] = round(mean("SQ_WAIT_ANY") / wc, 3) if mean("SQ_WAIT_ANY") else None
        e["active_inst_frac"] = round(mean("SQ_ACTIVE_INST_ANY") / wc, 3) if mean("SQ_ACTIVE_INST_ANY") else None
        if mean("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            e["mfma_busy_cycles"] = int(mean("SQ_VALU_MFMA_BUSY_CYCLES"))
        e["sq_wave_cycles"] = int(wc)
        e["sq_busy_cycles"] = int(busy)
        if mean("GRBM_GUI_ACTIVE"):
            e["grbm_gui_active"] = int(mean("GRBM_GUI_ACTIVE"))
    for k in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM"):
        if mean(k) is not None:
            e[k.lower()] = int(mean(k))
    for k in ("_VGPR_Count", "_Accum_VGPR_Count", "_LDS_Block_Size"):
        if c.get(k):
            e[k[1:].lower()] = int(c[k][0])
    out["kernels"].append(e)
json.dump(out, sys.stdout, indent=1)
