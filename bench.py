#!/usr/bin/env python3
"""bench.py -- sweeps/s of the 3D-Dual-Fusion hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run)

A "step" is one pass of the hot path over one batch of synthetic nuScenes-shaped sweeps that are
already resident in HBM: voxelize + mean VFE -> sparse 3-D backbone (21 fused sparse convs, 8
rulebooks) [-> dual-query deformable camera fusion] -> dense BEV [B,256,180,180].
Frames are independent: ranks process different sweeps, no data-path collective (weak scaling).

Besides the driver contract fields the JSON line carries
  roofline     : dominant kernel, algorithmic flops or bytes per launch / HIP-event time per launch
  cpu_baseline : the oracle (CPU restatement of the reference algorithm) timed on this box's host
                 cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "3d-dual-fusion_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_FP32_MFMA_TF = 157.3   # dense fp32 MFMA (= vector) peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=os.environ.get("DF3D_WORKLOAD", "auto"),
                    help="cp_fusion (BASELINE configs[1]) | cp_lidar (configs[0] shape) | auto")
    ap.add_argument("--batch", type=int, default=1, help="sweeps per GPU per step")
    ap.add_argument("--inflight", type=int, default=int(os.environ.get("DF3D_INFLIGHT", "1")),
                    help="frames in flight per GPU: each slot is a host thread + HIP stream + model replica; the K "
                         "timed steps are dealt to the slots (frames are independent).  1 (default) = strictly "
                         "sequential; 2 gains ~10 %% from K >= 40 steps on, nothing at K = 20")
    ap.add_argument("--conv-precision", default=os.environ.get("DF3D_CONV_PRECISION", "split"),
                    choices=["split", "fp32", "bf16"],
                    help="sparse-conv arithmetic: split (default, fp32-grade: bf16 hi+lo operands, 3 MFMA products), fp32 "
                         "(exact fp32 MFMA), bf16 (bf16 rows / weights, fp32 accumulate: BASELINE configs[2]-style, NOT the "
                         "fp32 configs[1] line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    return ap.parse_args()


def build_model(workload, dev):
    from dualfusion.pipeline import CenterPointHotPath
    torch.manual_seed(0)
    fusion = None
    if workload == "cp_fusion":
        from dualfusion.fusion import build_centerpoint_fusion
        fusion = build_centerpoint_fusion()
    model = CenterPointHotPath(fusion=fusion).eval().to(dev)
    # BatchNorm running statistics: mean 0 / var 1 defaults
    return model


def make_inputs(workload, batch, rank, dev):
    from dualfusion import synth
    pts = [torch.from_numpy(synth.nusc_sweep(seed=rank * 1000 + b)).to(dev) for b in range(batch)]
    extra = None
    if workload == "cp_fusion":
        from dualfusion.fusion import synthetic_camera_inputs
        extra = synthetic_camera_inputs(batch, dev, seed=1234 + rank)
    return pts, extra


def run_step(model, pts, extra):
    if extra is None:
        return model(pts)
    return model(pts, batch_dict=extra[0], example=extra[1])


PEAK_BF16_MFMA_TF = 2500.0  # dense bf16 MFMA peak; the split-precision kernels spend 3 bf16 products per fp32 product


def conv_algorithmic(rec, pairs):
    """SURVEY.md section 8(d) per launch: every valid (output, offset) pair reads one input row, every output row
    is written once (twice when the epilogue also emits the split rows the next layer gathers), plus the
    neighbour table and the filter bank:
        bytes = R*Cin*4 + N_out*Cout*4*(1 or 2) + K*N_out*4 + K*Cin*Cout*4 ;  flops = 2*R*Cin*Cout."""
    cin, cout, K, n_out = rec["cin"], rec["cout"], rec["kvol"], rec["n_out"]
    if rec["split"] == 2:      # bf16 kernel: 2-byte rows and weights; it writes bf16 rows and an fp32 copy of the result
        by = pairs * cin * 2 + n_out * cout * (2 + 4) + K * n_out * 4 + K * cin * cout * 2
    else:
        by = pairs * cin * 4 + n_out * cout * 4 * (2 if rec["split"] else 1) + K * n_out * 4 + K * cin * cout * 4
    return by, 2 * pairs * cin * cout


def roofline_from_timer(timer, meta_timer):
    """Dominant sparse-conv kernel of the timed region: HIP-event time per launch (timer) against the algorithmic
    bytes / flops of the same launches (pair counts from `meta_timer`, an extra untimed step with identical
    inputs).  Bound = whichever roof the kernel's arithmetic intensity puts it under: fp32 MFMA (157 TF) for the
    exact-fp32 kernels, bf16 MFMA / 3 for the split-precision kernels, HBM otherwise."""
    pairs_of = {}
    for r in meta_timer.records:
        pairs_of.setdefault((r["cin"], r["cout"], r["kvol"], r["n_out"]), []).append(r["pairs"])
    groups = {}
    for r in timer.records:
        key = (r["cin"], r["cout"], r["kvol"], r["split"])
        g = groups.setdefault(key, {"ms": 0.0, "n": 0, "by": 0, "fl": 0, "miss": 0})
        g["ms"] += r["ms"]
        g["n"] += 1
        cand = pairs_of.get((r["cin"], r["cout"], r["kvol"], r["n_out"]))
        if not cand:
            g["miss"] += 1
            continue
        # layers of one stage share the rulebook (same n_out -> same R); distinct tables of equal n_out average
        R = sum(cand) // len(cand)
        by, fl = conv_algorithmic(r, R)
        g["by"] += by
        g["fl"] += fl
    if not groups:
        return None, {}
    key = max(groups, key=lambda k: groups[k]["ms"])
    cin, cout, K, split = key
    g = groups[key]
    sec = g["ms"] * 1e-3
    counted = max(g["n"] - g["miss"], 1)
    by, fl = g["by"] * g["n"] // counted, g["fl"] * g["n"] // counted
    mfma_peak = PEAK_BF16_MFMA_TF if split == 2 else (PEAK_BF16_MFMA_TF / 3.0 if split else PEAK_FP32_MFMA_TF)
    ai = fl / max(by, 1)
    if ai >= mfma_peak * 1e12 / (PEAK_HBM_GBS * 1e9):
        ach = fl / sec / 1e12
        roof = {"bound": "mfma", "achieved": round(ach, 3), "peak": round(mfma_peak, 1), "unit": "TFLOP/s",
                "frac": round(ach / mfma_peak, 4)}
    else:
        ach = by / sec / 1e9
        roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(ach / PEAK_HBM_GBS, 4)}
    kname = ("spconv_os_split_kernel" if split else ("spconv_pair_kernel" if cout == 128 and cin >= 64
                                                     else "spconv_mfma_kernel"))
    roof.update({"traffic": None, "kernel": "%s<cin=%d,cout=%d,K=%d>" % (kname, cin, cout, K),
                 "precision": ("bf16 rows and weights, fp32 accumulate" if split == 2 else
                               "split bf16 hi/lo operands, 3 MFMA products, fp32 accumulate" if split else "fp32 MFMA"),
                 "launches": g["n"], "avg_launch_us": round(g["ms"] * 1e3 / g["n"], 2),
                 "algorithmic_flops_per_launch": fl // g["n"], "algorithmic_bytes_per_launch": by // g["n"],
                 "arithmetic_intensity_flop_per_byte": round(ai, 1)})
    # HBM traffic of that kernel comes from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; PMC cannot be
    # read from inside the process).  The committed summary of the last such run is attached when it belongs to
    # the same kernel; otherwise null.
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_spconv_split.json")))
        if pm.get("kernel_key") == [cin, cout, K, split]:
            roof["traffic"] = pm["traffic_bytes_per_launch"]
            roof["traffic_source"] = "profiles/r01_pmc_spconv_split.json (rocprofv3 --pmc, separate passes)"
    except Exception:
        pass
    per_kernel = {"%dx%d_k%d%s" % (k[0], k[1], k[2], "_split" if k[3] else ""):
                  {"ms_total": round(v["ms"], 3), "launches": v["n"]} for k, v in groups.items()}
    return roof, per_kernel


def cpu_baseline(workload, model, cam_np, budget_s=30.0):
    """The oracle composition (tests/oracle_models.py over oracle/oracle.py) of the SAME workload with the
    SAME weights on this box's host cores: voxelize -> VFE -> sparse backbone [-> projection, image gate,
    ACTR, write-back] -> dense.  Bounded: whole sweeps until ~budget_s have elapsed (at least one)."""
    import oracle_models as om
    from oracle import oracle as orc
    from dualfusion import synth
    from dualfusion.fusion import CP_DEPTH_THRES
    sd_all = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    sd = {k[len("backbone."):]: v for k, v in sd_all.items() if k.startswith("backbone.")}
    sd_f = {k[len("fusion."):]: v for k, v in sd_all.items() if k.startswith("fusion.")}
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        cores = os.cpu_count() or 1
    cores = max(cores, torch.get_num_threads())
    n, t0 = 0, time.perf_counter()
    while True:
        pts = synth.nusc_sweep(seed=n)
        t1 = time.perf_counter()
        ov, oc, on = orc.hard_voxelize(pts, synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, 120000, "numba")
        feats = orc.mean_vfe(ov, on)
        coors = np.concatenate([np.zeros((len(oc), 1), np.int32), oc], 1)
        fuse = None
        if workload == "cp_fusion":
            img, calib, hw = cam_np

            def fuse(c2, c3, c4):
                out = om.centerpoint_fusion(sd_f, [(c.indices, c.features) for c in (c2, c3, c4)], img, calib, hw,
                                            synth.NUSC_CAMS, synth.NUSC_VOXEL, synth.NUSC_RANGE, 2.0 / 3.0,
                                            CP_DEPTH_THRES)
                c4.features = out
                return c4
        om.centerpoint_backbone(sd, feats, coors, 1, [1440, 1440, 40], fuse=fuse)
        n += 1
        dt = time.perf_counter() - t1
        if time.perf_counter() - t0 + dt > budget_s or n >= 8:
            break
    total = time.perf_counter() - t0
    what = "voxelize+VFE+sparse backbone+camera fusion (projection, gate, ACTR)+dense" if workload == "cp_fusion" \
        else "voxelize+VFE+sparse backbone+dense, LiDAR branch"
    return {"value": round(n / total, 4), "unit": "sweeps/s", "cores": int(cores), "kind": "port",
            "sample": "%d whole synthetic sweeps (%s; C for index work, numpy/BLAS + torch-CPU fp32 for the dense "
                      "layers), %.1f s wall; host cpu_count=%d" % (n, what, total, os.cpu_count() or 0)}


def main():
    args = parse()
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback on the product path)"
    from dualfusion import dist as D
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rank, local, world = D.init_from_env("nccl")                # nccl == RCCL on ROCm (xGMI inside a node)
    workload = args.workload
    if workload == "auto":
        try:
            import dualfusion.fusion  # noqa: F401
            workload = "cp_fusion"
        except Exception:
            workload = "cp_lidar"
    from dualfusion import ops
    ops.CONV_PRECISION = args.conv_precision
    nslots = max(1, min(args.inflight, args.steps))
    # one slot = model replica (same seed -> same weights) + the same synthetic frame + its own HIP stream
    slots = []
    for _ in range(nslots):
        m = build_model(workload, dev)
        p, e = make_inputs(workload, args.batch, rank, dev)
        slots.append((m, p, e, torch.cuda.Stream(device=dev) if nslots > 1 else None))
    model, pts, extra, _ = slots[0]
    outs = [None] * nslots

    def barrier():
        D.barrier(dev)

    def work(i, n):
        m, p, e, st = slots[i]
        if st is None:
            for _ in range(n):
                outs[i] = run_step(m, p, e)
            return
        torch.cuda.set_device(dev)
        with torch.cuda.stream(st):
            for _ in range(n):
                outs[i] = run_step(m, p, e)

    for i in range(nslots):
        work(i, max(args.warmup, 1) if nslots > 1 else args.warmup)
    torch.cuda.synchronize()
    share = [args.steps // nslots + (1 if i < args.steps % nslots else 0) for i in range(nslots)]
    # Per-kernel HIP events live in the timed region only when one frame is in flight: with two, an event pair around
    # a launch also times whatever the other frame's stream co-runs, and the event packets serialise the queues
    # (measured: 580 -> 436 sweeps/s).  With frames in flight the roofline is therefore measured in a second pass of
    # the same K steps, sequentially, right after the timed region (reported with its own ms_per_step).
    # Events around EVERY conv launch cost the timed region ~6 % (30 launches per frame).  The roofline needs the
    # dominant kernel only: one untimed probe step with all events finds it (and yields the per-kernel table), the
    # timed region then records events around its launches alone.
    timer, probe = None, None
    if not args.no_kernel_timing:
        probe = ops.KernelTimer()
        probe.start()
        run_step(model, pts, extra)
        torch.cuda.synchronize()
        probe.stop()
        tot = {}
        for r in probe.records:
            k = (r["cin"], r["cout"], r["kvol"])
            tot[k] = tot.get(k, 0.0) + r["ms"]
        timer = ops.KernelTimer(only=max(tot, key=tot.get)) if tot else ops.KernelTimer()
    if nslots > 1:
        import threading
        threads = [threading.Thread(target=work, args=(i, share[i])) for i in range(nslots)]
    barrier()
    if timer is not None and nslots == 1:
        timer.start()
    t0 = time.perf_counter()
    if nslots == 1:
        work(0, args.steps)
    else:
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    barrier()                                  # synchronises the device (all streams) and the ranks
    elapsed = time.perf_counter() - t0
    out = outs[0]
    seq_elapsed = None
    if timer is not None and nslots > 1:
        timer.start()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            run_step(model, pts, extra)
        barrier()
        seq_elapsed = time.perf_counter() - t1
    if timer is not None:
        timer.stop()
        # metadata pass, outside the timed region: one more step with the same inputs that counts the valid
        # rulebook pairs of every conv launch (the unit the algorithmic bytes are stated in)
        meta_timer = ops.KernelTimer(count_pairs=True)
        meta_timer.start()
        run_step(model, pts, extra)
        torch.cuda.synchronize()
        meta_timer.stop()
    elapsed = D.max_over_ranks(elapsed, dev)
    dense = out[0]
    assert tuple(dense.shape) == (args.batch, 256, 180, 180), dense.shape
    if rank == 0:
        sweeps = args.steps * args.batch * world
        res = {
            "metric": "nuScenes sweeps/sec (0.075 m voxel, ~60k pts)", "value": round(sweeps / elapsed, 3),
            "unit": "sweeps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"split": "f32 (C>=64 sparse convs and the FFN: operands split into bf16 hi+lo, 3 MFMA products, fp32 "
                               "accumulate, ~1e-5 rel. error; everything else exact fp32)",
                      "fp32": "f32 (exact fp32 MFMA convolutions; FFN split precision)",
                      "bf16": "bf16 sparse convs (bf16 rows and weights, fp32 accumulate and epilogue); fusion adapter, "
                              "ACTR and C<=16 layers f32 -- NOT the fp32 configs[1] line"}[args.conv_precision],
            "data": "synthetic",
            "config": {"workload": {"cp_fusion": "CenterPoint + 3D-DF hot path (voxelize+VFE, SpMiddleResNetFHDFusion, "
                                                 "ACTR dual-query fusion on 6 synthetic DeepLabV3-shaped cam feats, dense BEV), "
                                                 "0.075 m voxel, fp32 [BASELINE configs[1]]",
                                    "cp_lidar": "CenterPoint voxelnet 0.075 m hot path, LiDAR branch only (voxelize+VFE, "
                                                "SpMiddleResNetFHD, dense BEV), fp32 [BASELINE configs[0] shape; camera fusion not in this line]"}[workload],
                       "frames_in_flight_per_gpu": nslots, "sweeps_per_gpu_per_step": args.batch, "points_per_sweep": int(pts[0].shape[0]),
                       "global_batch": args.batch * world, "parallelism": "dp%d (frames sharded, no data-path collective)" % world},
        }
        if timer is not None:
            roof, _ = roofline_from_timer(timer, meta_timer)
            _, per_kernel = roofline_from_timer(probe, meta_timer)        # all conv kernels, from the untimed probe step
            if roof is not None and seq_elapsed is not None:
                roof["measured_over"] = ("second pass of the same %d steps with ONE frame in flight (HIP events around "
                                         "every conv launch), %.4f ms/step; the timed region above keeps %d frames in "
                                         "flight without per-kernel events" % (args.steps, seq_elapsed / args.steps * 1e3,
                                                                               nslots))
                res["sequential_ms_per_step"] = round(seq_elapsed / args.steps * 1e3, 4)
            elif roof is not None:
                roof["measured_over"] = ("the timed region (one frame in flight; HIP events around the launches of this "
                                         "kernel only -- it was picked by an untimed probe step with events on every launch)")
            res["roofline"] = roof
            res["conv_kernel_ms_probe_step"] = per_kernel
        if world == 1 and not args.no_cpu_baseline:
            cam_np = None
            if workload == "cp_fusion":
                bd = extra[0]
                from dualfusion import synth
                img = {n: bd['img_feat']['layer1_ori_feat2d'][n.lower()].cpu().numpy() for n in synth.NUSC_CAMS}
                calib = {n: (bd['calib']['lidar2cam_' + n.lower().lstrip('cam_')].cpu().numpy(),
                             bd['calib']['cam_intrinsic_' + n.lower().lstrip('cam_')].cpu().numpy())
                         for n in synth.NUSC_CAMS}
                hw = tuple(int(v) for v in bd['image_shape']['cam_front'][0][:2])
                cam_np = (img, calib, hw)
            res["cpu_baseline"] = cpu_baseline(workload, model, cam_np)
        print(json.dumps(res))
    if D.is_dist():
        D.barrier(dev)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
