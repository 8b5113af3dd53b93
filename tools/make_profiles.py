#!/usr/bin/env python3
"""Distil gpurun_out/prof_r02 (written by tools/profile_r02.sh on an MI355X box) into profiles/<tag>_*.
usage: make_profiles.py <gpurun_out/prof_r02> <round tag, e.g. r02>"""
import collections
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1].rstrip("/") + "/", sys.argv[2]
out = os.path.join(ROOT, "profiles") + "/"
N_CAL = 1 << 21


def rows(fn):
    return list(csv.DictReader(open(src + fn)))


def mean(rs, cname, pred):
    vals = [float(r["Counter_Value"]) for r in rs if r["Counter_Name"] == cname and pred(r)]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)


bench = json.loads([l for l in open(src + "bench.json").read().strip().splitlines() if l.startswith("{")][-1])
roof = bench["roofline"]


def wgs(r):
    return int(r["Grid_Size"]) // int(r["Workgroup_Size"])


def selector(cin, cout):
    """The K = 27 launches of the cin -> cout split-precision kernel: by template arguments and tile count (the
    neck / head launch the same kernels on the dense 180 x 180 / 90 x 90 maps with other tile counts)."""
    tiles = {(n + 127) // 128 for n in bench["conv_rows_by_kernel"]["%dx%d_k27" % (cin, cout)]}
    pat = "<%d, %d" % (cin, cout)
    return lambda r: "spconv_os" in r["Kernel_Name"] and pat in r["Kernel_Name"] and wgs(r) in tiles


cal = lambda r: "spconv_os" in r["Kernel_Name"] and wgs(r) == N_CAL // 128    # noqa: E731
fetch, write = rows("fetch/fetch_counter_collection.csv"), rows("write/write_counter_collection.csv")
cf, ncf = mean(rows("calib_fetch/fetch_counter_collection.csv"), "FETCH_SIZE", cal)
cw, ncw = mean(rows("calib_write/write_counter_collection.csv"), "WRITE_SIZE", cal)
exp_f, exp_w = (N_CAL * 512 + N_CAL * 4) / 1024.0, 2 * N_CAL * 512 / 1024.0
kf, kw = exp_f / cf, exp_w / cw
calib = {"what": "K = 1 'convolution' 128 -> 128 over 2^21 rows, neighbour table = random permutation: every 512-byte split row "
                 "of the 1 GiB input is gathered exactly once by the conv kernel itself (tools/ubench/pmc_calib.py)",
         "expected_fetch_KB": exp_f, "FETCH_SIZE_KB_mean": cf, "fetch_factor": round(kf, 4),
         "expected_write_KB": exp_w, "WRITE_SIZE_KB_mean": cw, "write_factor": round(kw, 4), "launches": ncf}
json.dump(calib, open(out + tag + "_pmc_calibration.json", "w"), indent=1)
trace = rows("trace/stats_kernel_trace.csv")


def summarize(cin, cout):
    sel = selector(cin, cout)
    f, nf = mean(fetch, "FETCH_SIZE", sel)
    w, nw = mean(write, "WRITE_SIZE", sel)
    tiles = {(n + 127) // 128 for n in bench["conv_rows_by_kernel"]["%dx%d_k27" % (cin, cout)]}
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in trace
           if "spconv_os" in r["Kernel_Name"] and "<%d, %d" % (cin, cout) in r["Kernel_Name"]
           and int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) in tiles and int(r["Grid_Size_Y"]) == 1]
    names = sorted({r["Kernel_Name"].split("(")[0] for r in fetch if sel(r)})
    traffic = int((kf * f + kw * w) * 1024)
    return {"kernel": "%s launched with K = 27 (3-D SubM layers; the dense-map launches of the same kernel are excluded by their "
                      "tile count)" % " / ".join(names),
            "kernel_key": [cin, cout, 27, 1], "FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w, "launches_averaged": nf,
            "traffic_bytes_per_launch": traffic, "fetch_bytes_per_launch": int(kf * f * 1024),
            "write_bytes_per_launch": int(kw * w * 1024), "rocprofv3_avg_launch_us": round(sum(dur) / len(dur), 2),
            "hbm_rate_GBps_over_rocprof_time": round(traffic / (sum(dur) / len(dur) * 1e-6) / 1e9, 1)}, sel


import re
m = re.search(r"cin=(\d+),cout=(\d+),K=(\d+)", roof["kernel"]) or re.search(r"<(\d+), (\d+)[^>]*> K=(\d+)", roof["kernel"])
dom = (int(m.group(1)), int(m.group(2)))
pm, conv_sel = summarize(*dom)
pm.update({"fetch_calibration": {"factor": round(kf, 4), "write_factor": round(kw, 4),
                                 "source": "profiles/%s_pmc_calibration.json (known-size gather by the conv kernel)" % tag},
           "correction": "FETCH_SIZE x %.3f, WRITE_SIZE x %.3f (calibrated on a known-size gather in this kernel's access pattern, "
                         "MI355X_MICROARCH.md HBM section); x1024 (values are KB)" % (kf, kw),
           "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"], "compulsory_bytes": roof.get("compulsory_bytes"),
           "bench_hip_event_avg_launch_us": roof["avg_launch_us"],
           "command": "tools/profile_%s.sh: rocprofv3 --pmc FETCH_SIZE (then WRITE_SIZE, separate passes) --kernel-trace "
                      "--output-format csv -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-passes "
                      "--no-kernel-timing (8 rotating frames)" % ("r02" if tag == "r02" else tag + "_pmc")})
others = {}
for key in bench["conv_rows_by_kernel"]:
    cin, cout = [int(v) for v in key.split("_")[0].split("x")]
    if (cin, cout) != dom and cin >= 32:
        try:
            others[key] = summarize(cin, cout)[0]
            others[key]["probe_step"] = bench.get("roofline_probe_step_by_kernel", {}).get(key)
        except Exception as e:
            print("no summary for", key, e)
pm["other_k27_kernels"] = others
json.dump(pm, open(out + tag + "_pmc_spconv_split.json", "w"), indent=1)
shutil.copy(src + "trace/stats_kernel_stats.csv", out + tag + "_bench_cp_fusion_kernel_stats.csv")
json.dump(bench, open(out + tag + "_bench_cp_fusion.json", "w"), indent=1)
for fn, o, cn in (("fetch/fetch_counter_collection.csv", "_pmc_fetch_size.csv", "FETCH_SIZE"),
                  ("write/write_counter_collection.csv", "_pmc_write_size.csv", "WRITE_SIZE")):
    agg = collections.OrderedDict()
    for r in rows(fn):
        if r["Counter_Name"] != cn or not ("spconv" in r["Kernel_Name"] or "ffn_split" in r["Kernel_Name"] or "img_proj" in r["Kernel_Name"]
                                           or "msda" in r["Kernel_Name"]):
            continue
        k = (r["Kernel_Name"].split("(")[0][:90], r["Grid_Size"])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    with open(out + tag + o, "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(["Kernel_Name", "Grid_Size", "Launches", cn + "_KB_mean"])
        for (k, g), (n, s) in agg.items():
            wr.writerow([k, g, n, "%.1f" % (s / n)])
# SQ counters of the dominant kernel
try:
    sq = rows("sq/sq_counter_collection.csv")
    names = sorted(set(r["Counter_Name"] for r in sq))
    doc = {"note": "SQ_* counters count quad-cycles per SIMD-wave slot except SQ_VALU_MFMA_BUSY_CYCLES (cycles); "
                   "GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / kernel time = effective clock", "kernels": {}}
    for key in bench["conv_rows_by_kernel"]:
        cin, cout = [int(v) for v in key.split("_")[0].split("x")]
        if cin < 32:
            continue
        sel = selector(cin, cout)
        doc["kernels"][key] = {"kernel": " / ".join(sorted({r["Kernel_Name"].split("(")[0] for r in sq if sel(r)})),
                               "mean_per_launch": {c: mean(sq, c, sel)[0] for c in names}}
    json.dump(doc, open(out + tag + "_pmc_sq_spconv.json", "w"), indent=1)
except Exception as e:      # the SQ pass is optional
    print("no SQ summary:", e)
# round 3: WRITE_SIZE of the sparse-conv kernels with and without the fp32 copies nobody reads (DF3D_EXEC_F32_ALL=1 = round 2)
if os.path.exists(src + "write_all/write_counter_collection.csv"):
    def totals(fn):
        agg = collections.OrderedDict()
        for r in rows(fn):
            if r["Counter_Name"] == "WRITE_SIZE" and "spconv" in r["Kernel_Name"] and int(r["Grid_Size"]) // int(r["Workgroup_Size"]) > 0:
                k = r["Kernel_Name"].split("(")[0][:80]
                a = agg.setdefault(k, [0, 0.0])
                a[0] += 1
                a[1] += float(r["Counter_Value"])
        return agg
    now, before = totals("write/write_counter_collection.csv"), totals("write_all/write_counter_collection.csv")
    doc = {"what": "WRITE_SIZE (KB, x %.3f calibration) summed over ALL launches of every sparse-conv kernel in the same bench "
                   "command (tools/profile_%s_pmc.sh), default build vs DF3D_EXEC_F32_ALL=1 (every layer writes fp32 AND split "
                   "rows, as in round 2).  Since round 3 a split-precision layer of the native executor writes fp32 rows only "
                   "when they are read: exported stages, residual sources, inputs of non-split layers." % (kw, tag),
           "kernels": {}}
    tn = tb = 0.0
    for k in before:
        n = now.get(k, [0, 0.0])
        doc["kernels"][k] = {"launches": before[k][0], "write_KB_round2_behaviour": round(before[k][1] * kw, 1),
                             "write_KB_now": round(n[1] * kw, 1), "launches_now": n[0]}
        tn += n[1] * kw
        tb += before[k][1] * kw
    doc["total_write_KB_round2_behaviour"], doc["total_write_KB_now"] = round(tb, 1), round(tn, 1)
    doc["saved_fraction"] = round(1 - tn / tb, 4) if tb else None
    json.dump(doc, open(out + tag + "_pmc_write_ab.json", "w"), indent=1)
    print(json.dumps({k: doc[k] for k in ("total_write_KB_round2_behaviour", "total_write_KB_now", "saved_fraction")}))
ps = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_summary.py"), src + "trace/stats_kernel_trace.csv", "--csv",
                     out + tag + "_bench_cp_fusion_per_step.csv"], capture_output=True, text=True)
print(ps.stdout[:3000])
print(json.dumps(pm, indent=1))
print(json.dumps(calib, indent=1))
