#!/usr/bin/env python3
"""DESIGN.md section 3.1 from the committed profile of a round: one row per kernel class of the headline step
(profiles/<tag>_bench_cp_fusion.json `roofline_by_kernel` = HIP-event time per step; rocprofv3 per-step table for the in-step
kernel time; FETCH / WRITE / SQ counter summaries for the "counter that proves it" column).
usage: design_kernel_table.py r06 > table.md"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = lambda n: os.path.join(ROOT, "profiles", "%s_%s" % (tag, n))   # noqa: E731
bench = json.load(open(P("bench_cp_fusion.json")))
pmc = json.load(open(P("pmc_spconv_split.json")))
sq = json.load(open(P("pmc_sq_spconv.json")))["kernels"]
step = {}
for r in csv.DictReader(open(P("bench_cp_fusion_per_step.csv"))):
    step[r["Name"]] = (float(r["CallsPerStep"]), float(r["UsPerStep"]), float(r["AvgUs"]))

LAYERS = {"rulebook": "frame head: occupancy directories, 3 strided output index sets, 8 neighbour tables (side stream, a frame ahead)",
          "conv 128->128 K=27": "conv4 SubM blocks (4 layers)", "conv 64->64 K=27": "conv3 SubM blocks (4)",
          "conv 32->32 K=27": "conv2 SubM blocks (4)", "conv 16->16 K=27": "conv1 SubM blocks (4)",
          "conv 32->64 K=27": "conv3 strided", "conv 64->128 K=27": "conv4 strided", "conv 64->2304 K=9": "CenterHead: 36 first convs as one grouped launch",
          "conv 512->64 K=9": "CenterHead shared conv", "conv 256->256 K=9": "RPN block 2 (90 x 90 map, 5 layers)",
          "conv 128->128 K=9": "RPN block 1 (180 x 180, 5 layers)", "conv 256->128 K=9": "RPN block 1 entry (ZeroPad + conv)",
          "conv 256->256 K=4": "RPN deblock 2 (transposed, k = s = 2)", "conv 128->256 K=9": "RPN block 2 entry (stride 2)",
          "df3d_ffn_fused_jobs": "both feed-forward blocks + LayerNorm of a dual-query layer (2 layers)",
          "df3d_ms_deform_attn_fused": "deformable sampling of the 6 x max_ne queries (2 layers)",
          "df3d_rows_linear": "offset / weight linears, output projection + LayerNorm, image-query projection",
          "df3d_value_fold_gemm": "GroupNorm fold + value rows of both layers from the projected camera maps",
          "df3d_imgproj_split": "input projection + gate summary of the 6 camera maps (one pass over 246 MB)",
          "df3d_head_final_conv_packed": "CenterHead: 36 final 3 x 3 convs", "df3d_hard_voxelize_batched": "voxelisation + mean VFE (side stream)",
          "df3d_assemble_queries2_slots": "per-camera query tensors", "df3d_bigate_sum": "bidirectional gate (2 layers)",
          "df3d_gate_scatter_rows": "image gate: voxel side", "df3d_query_slots": "query slots per camera", "df3d_conv_pack_weights": None}


def counter(k):
    key = k.replace("conv ", "").replace("->", "x").replace(" K=", "_k")
    ent = pmc if "x".join(str(v) for v in pmc.get("kernel_key", [])[:2]) + "_k%d" % pmc.get("kernel_key", [0, 0, 0])[2] == key else \
        pmc.get("other_k27_kernels", {}).get(key)
    out = []
    if ent:
        alg = ent.get("algorithmic_bytes_per_launch") or ent.get("probe_step", {}).get("algorithmic_bytes_per_launch")
        out.append("FETCH + WRITE %.0f MB per launch%s, HBM at %.2f TB/s" % (
            ent["traffic_bytes_per_launch"] / 1e6, (" = %.2f x algorithmic" % (ent["traffic_bytes_per_launch"] / alg)) if alg else "",
            ent["hbm_rate_GBps_over_rocprof_time"] / 1e3))
    s = sq.get(key)
    if s:
        m = s["mean_per_launch"]
        out.append("waves waiting %.0f %% of their cycles (SQ_WAIT_ANY / SQ_WAVE_CYCLES), issuing %.0f %%"
                   % (100.0 * m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 100.0 * m["SQ_ACTIVE_INST_ANY"] / m["SQ_WAVE_CYCLES"]))
    return "; ".join(out)


print("| kernel class (entry) | layers it serves | launches / step | us / step (HIP events) | fraction of its roof | bound | counter evidence (`profiles/%s_pmc_*`) |" % tag)
print("|---|---|---|---|---|---|---|")
tot = 0.0
for r in bench["roofline_by_kernel"]:
    if LAYERS.get(r["kernel"], "") is None:
        continue                                    # (weight re-packing of the probe's precision switch: not part of a steady step)
    tot += r["us_per_step"]
    frac = "%.3f" % r["frac"] if "frac" in r else "-"
    roof = "%s %s" % (r.get("achieved", ""), r.get("unit", "")) if "achieved" in r else ""
    print("| `%s` | %s | %d | %.1f | %s%s | %s | %s |" % (r["kernel"], LAYERS.get(r["kernel"], r.get("what") or ""), r["launches"], r["us_per_step"],
                                                     frac, (" (%s)" % roof) if roof else "", r.get("bound", "-"), counter(r["kernel"])))
print()
print("Sum of the rows: %.2f ms of kernel time per step (every class >= 30 us); the step itself: %.3f ms (`ms_per_step`), hot path %.3f ms."
      % (tot / 1e3, bench["ms_per_step"], bench["hot_path"]["ms_per_step"]))
