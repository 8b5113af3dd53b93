#!/usr/bin/env python3
"""bench.py -- sweeps/s of the 3D-Dual-Fusion hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment the script starts its own N ranks (one process per GPU,
torch.distributed.run, 127.0.0.1 rendezvous, backend nccl = RCCL over xGMI); launched BY torch.distributed.run it
reads RANK / LOCAL_RANK / WORLD_SIZE from the environment.  Either way rank 0 prints ONE JSON line.

A "step" (default workload cp_fusion = BASELINE configs[1]) is one pass of the detector's forward over one batch of
synthetic nuScenes-shaped sweeps resident in HBM, the same at every N:
    points -> voxelize + mean VFE -> sparse 3-D backbone (21 fused sparse convs, 8 rulebooks) -> dual-query deformable
    camera fusion (ACTR) -> dense BEV -> RPN neck -> CenterHead -> detection losses (device) -> reduce_dict over the
    ranks (RCCL reduce of the loss scalars; a no-op at N = 1).
Frames are independent: ranks process different sweeps (weak scaling); the loss reduction is the path's only
collective.  Every step takes the NEXT of >= 8 distinct frames (sweep, camera feature maps, calibration, targets) and
fresh batch_dict / example objects, as a data loader would hand them over.

Besides the driver contract the JSON line carries
  hot_path      the same K steps ending at the dense BEV tensor (round-1's step), timed in a second pass
  fp32          both of the above with every sparse conv on the exact-fp32 MFMA kernels (--conv-precision fp32)
  roofline      dominant sparse-conv kernel: SURVEY 8(d) algorithmic bytes / HIP-event time, measured in the timed region
  cpu_baseline  the reference's own compiled CPU ops (oracle/_ref) on this box's host cores (rank 0, N = 1 only)
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

# HIP maps a process's streams onto at most GPU_MAX_HW_QUEUES hardware queues (default 4); streams that share a queue run their
# kernels one after the other.  A detector step here uses ~8 streams (frame-head worker, geometry, voxeliser, adapter side streams,
# RCCL) and the in_flight pass two detectors: with 4 queues the two frames in flight did not overlap at all (2.95 ms per step
# against 2.40 with 16, profiles/r04_*).  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
# Kernel arguments in DEVICE memory instead of host-coherent memory: a launch's argument block is then read from HBM, not over
# PCIe -- with ~130 dependent launches per frame, many of them 5-20 us long, that is 125 us of a 2.9 ms step (same-box A/B,
# tools/ab.sh HIP_FORCE_DEV_KERNARG 0 1: 2.90 -> 2.77 ms).  A runtime setting AMD documents for MI300-class parts; before HIP init.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_HBM_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_FP32_MFMA_TF = 157.3   # dense fp32 MFMA (= vector) peak
PEAK_BF16_MFMA_TF = 2500.0  # dense bf16 / fp16 MFMA peak; the split-precision kernels spend 3 products per fp32 product
PMC_SUMMARY = os.path.join(ROOT, "profiles", "r06_pmc_spconv_split.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default=os.environ.get("DF3D_WORKLOAD", "cp_fusion"),
                    choices=["cp_fusion", "cp_lidar", "tf_fusion", "vr_fusion", "protocol"],
                    help="cp_fusion = BASELINE configs[1] (default, the headline); cp_lidar = configs[0] shape; tf_fusion = "
                         "configs[2] (TransFusion-L + 3D-DF, bs=4, bf16 convs); vr_fusion = configs[4] (Voxel-RCNN + 3D-DF, "
                         "KITTI, bs=8); protocol = launcher / barrier / reduce protocol only, no GPU work (CPU tests)")
    ap.add_argument("--stage", default="detect", choices=["detect", "hot_path", "train", "boxes"],
                    help="detect (default): ... -> neck -> head -> losses -> reduce_dict; hot_path: stop at the dense BEV; "
                         "train (cp_lidar, cp_fusion, tf_fusion): forward + backward + bucketed gradient all-reduce overlapped with backward "
                         "+ AdamW step + reduce_dict of the losses -- a real data-parallel training step; boxes (tf_fusion): "
                         "... -> head -> decoded boxes instead of the losses")
    ap.add_argument("--batch", type=int, default=0, help="sweeps per GPU per step (0 = the workload's BASELINE batch)")
    ap.add_argument("--frames", type=int, default=8, help="distinct synthetic frames the steps rotate through")
    ap.add_argument("--inflight", type=int, default=2,
                    help="extra pass at N = 1: the same steps with this many frames in flight in ONE process -- detector "
                         "replicas, one HIP stream each, all queued by one host thread; every replica's count round trips are "
                         "taken a frame ahead by its native frame-head worker.  Reported as `in_flight` (with per-frame latency), "
                         "never as `value`.  1 = skip the pass")
    ap.add_argument("--conv-precision", default=os.environ.get("DF3D_CONV_PRECISION", ""),
                    choices=["", "split", "split3", "fp32", "bf16"],
                    help="sparse-conv arithmetic: split (fp32-grade: fp16 hi+lo operands, 3 MFMA products), fp32 (exact fp32 "
                         "MFMA), bf16 (bf16 rows / weights, fp32 accumulate).  Default: split for the fp32 configs, bf16 "
                         "for tf_fusion (configs[2] is a bf16 config)")
    ap.add_argument("--backend", default="", help="torch.distributed backend (default nccl = RCCL; gloo for the CPU tests)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-extra-passes", action="store_true", help="skip the hot_path / fp32 passes after the timed region")
    ap.add_argument("--skip-passes", default=os.environ.get("DF3D_BENCH_SKIP", ""),
                    help="comma list of extra passes to leave out: hot_path, in_flight, split3, fp32 (debugging aid)")
    ap.add_argument("--no-prefetch", dest="prefetch", action="store_false",
                    help="cp_fusion / cp_lidar: do NOT start the next frame's voxelisation / rulebooks / query slots on the "
                         "detector's helper thread while the current frame is queued (round 3's behaviour: every count round "
                         "trip on the queueing thread)")
    ap.add_argument("--no-tape", dest="tape", action="store_false",
                    help="cp_fusion / cp_lidar: run neck + head through the module path every frame instead of re-issuing their "
                         "recorded launches (dualfusion/tape.py; same kernels, same results, ~0.3 ms less host time per frame)")
    ap.add_argument("--no-side-configs", dest="side_configs", action="store_false",
                    help="skip the ~10-step passes over BASELINE configs[0] / [2] / [4] after the headline (N = 1 only)")
    ap.add_argument("--cpu-sweeps", type=int, default=20,
                    help="timed sweeps of the CPU baseline at the fastest thread setting (SURVEY 8(d): 20 after 3 untimed ones)")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------ workloads
class ProtocolWorkload(object):
    """No GPU work: exercises exactly the launcher, the barrier / max-over-ranks timing and the loss reduction of this
    script with any backend (tests/test_dist_gloo.py drives it with gloo on CPU)."""
    name, unit_name, batch = "protocol", "sweeps", 1

    def __init__(self, args, rank, world, dev):
        self.rank, self.dev = rank, dev

    def describe(self):
        return "protocol check: no device work, loss scalars = f(rank) reduced over the ranks"

    def step(self, i, stage):
        time.sleep(0.002)
        return {"loss": torch.tensor([1.0 + self.rank], device=self.dev),
                "hm_loss": torch.tensor([10.0 * (1 + self.rank)], device=self.dev)}

    def check(self, out, stage):
        pass


class CenterPointWorkload(object):
    """BASELINE configs[1] (cp_fusion) / configs[0] shape (cp_lidar): CenterPoint voxelnet 0.075 m [+ 3D-DF fusion on six
    DeepLabV3-shaped camera feature maps], RPN neck, 6-task CenterHead, device losses."""
    unit_name = "sweeps"

    def __init__(self, args, rank, world, dev):
        from dualfusion import synth
        from dualfusion.pipeline import NUSC_TASKS, CenterPointDetector
        self.name = args.workload
        self.batch = args.batch or 1
        self.dev, self.rank = dev, rank
        torch.manual_seed(0)                          # every rank holds the same replica (BN: mean 0 / var 1 defaults)
        fusion = None
        if self.name == "cp_fusion":
            from dualfusion.fusion import build_centerpoint_fusion
            fusion = build_centerpoint_fusion()
        self.model = CenterPointDetector(fusion=fusion).eval().to(dev)
        # the sweeps are resident inputs, complete before the timed region: voxelisation may run on its own stream
        self.model.hot_path.resident_inputs = os.environ.get("DF3D_VOXEL_STREAM", "1") == "1"
        if fusion is not None:
            fusion.resident_inputs = self.model.hot_path.resident_inputs
        self.num_classes = [t["num_class"] for t in NUSC_TASKS]
        self.prefetch = args.prefetch and self.model.hot_path.resident_inputs
        self.model.launch_tape = bool(getattr(args, "tape", True))
        self.stride, self._staged = 1, {}
        self.frames = []
        for f in range(max(1, args.frames)):
            seed = rank * 1000 + f
            fr = {"points": [torch.from_numpy(synth.nusc_sweep(seed=seed * 16 + b)).to(dev) for b in range(self.batch)]}
            tg = synth.centerhead_targets(self.batch, self.num_classes, seed=seed)
            fr["targets"] = {k: [torch.from_numpy(a).to(dev) for a in v] for k, v in tg.items()}
            if fusion is not None:
                fr["cam"] = self._camera_frame(seed)
            self.frames.append(fr)

    def _camera_frame(self, seed, raw_hw=(900, 1600), image_scale=2.0 / 3.0, feat_hw=(150, 267)):
        """Camera-side inputs of ONE frame: the 2-D network's output for its B*6 images as one tensor, the per-sample
        calibration (the rig yaw differs from frame to frame, as nuScenes' lidar2cam does) and the image shapes."""
        from dualfusion import synth
        B = self.batch
        feats = torch.from_numpy(synth.camera_features(B * 6, 256, feat_hw, 1234 + seed).reshape(
            B, 6, 256, feat_hw[0], feat_hw[1])).to(self.dev)
        cams = synth.nusc_cameras(image_hw=raw_hw, yaw_offset_deg=0.37 * (seed % 97))
        H, W = int(round(raw_hw[0] * image_scale)), int(round(raw_hw[1] * image_scale))
        calib = {}
        for name in synth.NUSC_CAMS:
            T, K = cams[name]
            ck = name.lower().lstrip('cam_')
            calib['lidar2cam_' + ck] = torch.from_numpy(np.stack([T] * B)).to(self.dev)
            calib['cam_intrinsic_' + ck] = torch.from_numpy(np.stack([K] * B)).to(self.dev)
        return {"feats": feats, "calib": calib, "shape": [H, W, 3]}

    def fresh_inputs(self, fr):
        """New batch_dict / example OBJECTS around the frame's resident tensors -- what a data loader hands over per
        iteration (nothing downstream may key a cache on the identity of these dicts)."""
        from dualfusion import synth
        example = {k: list(v) for k, v in fr["targets"].items()}
        if "cam" not in fr:
            return None, example
        cam = fr["cam"]
        bd = {'image_shape': {}, 'img_feat': {'layer1_ori_feat2d': {}}, 'calib': dict(cam["calib"])}
        for i, name in enumerate(synth.NUSC_CAMS):
            key = name.lower()
            bd['image_shape'][key] = torch.tensor([cam["shape"]] * self.batch)
            bd['img_feat']['layer1_ori_feat2d'][key] = cam["feats"][:, i]
        return bd, example

    def describe(self):
        return {"cp_fusion": "CenterPoint + 3D-DF detector forward (voxelize+VFE, SpMiddleResNetFHDFusion, ACTR dual-query "
                             "fusion on 6 synthetic DeepLabV3-shaped cam feats, dense BEV, RPN neck, CenterHead, detection "
                             "losses), 0.075 m voxel, fp32 [BASELINE configs[1]]",
                "cp_lidar": "CenterPoint voxelnet 0.075 m detector forward, LiDAR branch only (voxelize+VFE, "
                            "SpMiddleResNetFHD, dense BEV, RPN neck, CenterHead, detection losses), fp32 [BASELINE "
                            "configs[0] shape; camera fusion not in this line]"}[self.name]

    def _train_setup(self):
        from dualfusion import dist as D
        self.model.train()
        params = [p for p in self.model.parameters() if p.requires_grad]
        self.reducer = D.GradBucketReducer(params)      # 16 MB buckets: three for this detector
        if os.environ.get("DF3D_BUCKET_ADAMW", "1") == "1":
            self.optimizer = D.BucketAdamW(self.reducer, lr=1e-4, weight_decay=0.01)   # one fused launch per 16 MB bucket
        else:
            self.optimizer = torch.optim.AdamW(params, lr=1e-4, weight_decay=0.01, fused=True)
        self.n_params = sum(p.numel() for p in params)

    def close(self):
        self.model.close()

    def step(self, i, stage):
        fr = self.frames[i % len(self.frames)]
        staged = self._staged.pop(i, None) if stage != "train" else None
        bd, example = staged if staged is not None else self.fresh_inputs(fr)
        if self.prefetch and stage != "train":
            # the data loader's next batch: its voxelisation, rulebooks and query slots (everything that depends on the raw
            # inputs alone, with the frame's count round trips) start on the detector's helper thread while THIS frame is
            # queued (dualfusion/prefetch.py; the reference voxelises the next batch in its DataLoader workers the same way).
            # `stride`: with several frames in flight this replica sees every stride-th frame.
            nxt = i + self.stride
            frn = self.frames[nxt % len(self.frames)]
            self._staged = {nxt: self.fresh_inputs(frn)}
            self.model.prefetch(frn["points"], self._staged[nxt][0])
        if stage == "train":
            if getattr(self, "reducer", None) is None:
                self._train_setup()
            self.reducer.zero_grad()
            # logging copies without a host wait (a trainer reads them every N steps, behind `host_copies_ready`): the host
            # queues the optimizer step and the next frame while the backward still runs
            rets = self.model.training_step(fr["points"], example, batch_dict=bd,
                                            host_copies="async" if os.environ.get("DF3D_TRAIN_ASYNC_LOG", "1") == "1" else True)
            self.reducer.finish()                      # waits for the bucket all-reduces launched during backward
            self.optimizer.step()
            dev_side = rets.get("on_device", {})           # async logging: the pinned host copies are not read back here
            return {k: torch.stack([v.detach().to(self.dev).float().reshape(()) for v in dev_side.get(k, rets[k])])
                    for k in ("loss", "hm_loss", "loc_loss")}
        if stage == "hot_path":
            hp = self.model.hot_path
            neck, hp.neck, hp.backbone.dense_layout = hp.neck, None, "nchw"
            try:
                return hp(fr["points"], batch_dict=bd, example=example)[0]
            finally:
                hp.neck, hp.backbone.dense_layout = neck, "rows"
        return self.model(fr["points"], batch_dict=bd, example=example, return_loss=True)

    def check(self, out, stage):
        if stage == "hot_path":
            assert tuple(out.shape) == (self.batch, 256, 180, 180), out.shape
        elif stage == "train":
            assert bool(torch.isfinite(out["loss"]).all()), out
        else:
            v = out["loss"].float().cpu()
            assert v.shape == (6,) and bool(torch.isfinite(v).all()), v


def make_workload(args, rank, world, dev):
    if args.workload == "protocol":
        return ProtocolWorkload(args, rank, world, dev)
    if args.workload in ("cp_fusion", "cp_lidar"):
        return CenterPointWorkload(args, rank, world, dev)
    from dualfusion.workloads import make as make_tree_workload      # tf_fusion / vr_fusion (configs[2] / [4])
    return make_tree_workload(args, rank, world, dev)


# ------------------------------------------------------------------------------------------------ roofline
def conv_algorithmic(rec, pairs):
    """SURVEY.md section 8(d), per launch:  bytes = R*Cin*s + N_out*Cout*s + 8*R + K*Cin*Cout*s ;  flops = 2*R*Cin*Cout
    (R = valid rulebook pairs, s = bytes per element of the rows: 4, or 2 for the bf16 kernels).  Every output row is
    counted ONCE: the second copy the split-precision epilogue writes when BOTH the caller / a residual add (fp32 rows) and
    the next layer (hi/lo bf16 rows) read the output is an implementation choice, reported separately as
    `extra_written_bytes_per_launch` (round 3: only the launches that really write both -- the first conv of a residual
    block writes split rows only)."""
    cin, cout, K, n_out = rec["cin"], rec["cout"], rec["kvol"], rec["n_out"]
    s = 2 if rec["split"] == 2 else 4
    if K in (1, 4, 9) and pairs >= 0.9 * K * n_out:
        # a DENSE 2-D map (BEV neck, detection heads: the sparse kernel over a full neighbour table).  The sparse formula would
        # count every pixel K times; a dense convolution's algorithmic traffic is every input pixel once, every output pixel
        # once and the filters -- with it these launches fall under the matrix roof, where they belong
        by = (pairs // K) * cin * s + n_out * cout * s + K * cin * cout * s
        return by, 2 * pairs * cin * cout, n_out * cout * 4 if rec.get("both") else 0
    by = pairs * cin * s + n_out * cout * s + 8 * pairs + K * cin * cout * s
    extra = n_out * cout * 4 if rec.get("both") else 0
    return by, 2 * pairs * cin * cout, extra


def trace_kernel_name(cin, cout, K, split, n_out):
    """Name of the kernel libdf3d_hip.so launches for this shape, as a rocprofv3 kernel trace prints it (so that this table
    joins `profiles/*_kernel_stats.csv`): mirrors the dispatch in csrc/spconv.hip (`dispatch_small`, `dispatch_cin`) and
    csrc/spconv_split.hip (`launch_os_split`, `launch_os_split_wide`, `use_lc`)."""
    if not split:
        if n_out >= 2048 and (cin, cout) in ((16, 16), (5, 16), (4, 16), (16, 32)):
            if os.environ.get("DF3D_ROW_LISTS", "1") != "0" and os.environ.get("DF3D_EXECUTOR", "1") != "0":
                # the executor builds row lists for these tables (round 4): `dispatch_small_lists`
                return "spconv_small_lists_kernel<%d, %d, %d>" % (cin, cout, 512 if cout == 32 else 256)
            return "spconv_small_kernel<%d, %d>" % (cin, cout)
        return "spconv_mfma_kernel<%d, %d, ...>" % (max(cin, 8), cout)
    gy = max(1, cout // 128) if cout % 128 == 0 else 1
    if cout % 128 == 0 and cin % 32 == 0 and K > 1 and -(-n_out // 128) * gy >= 190:
        return "spconv_os_lc_kernel<%d, 128>" % cin
    return "spconv_os_split_kernel<%d, %d, ...>" % (cin, cout)


def roofline_from_timer(timer, meta_timer, want=None):
    """Dominant sparse-conv kernel of the timed region: HIP-event time per launch (`timer`) against the algorithmic
    bytes / flops of the same launches (pair counts from `meta_timer`: one extra untimed pass over every frame).
    Bound = whichever roof the kernel's arithmetic intensity puts it under: fp32 MFMA (157 TF) for the exact-fp32
    kernels, bf16 MFMA / 3 for the split-precision kernels, HBM otherwise."""
    pairs_of = {}
    for r in meta_timer.records:
        pairs_of.setdefault((r["cin"], r["cout"], r["kvol"], r["n_out"]), []).append(r["pairs"])
    groups = {}
    for r in timer.records:
        key = (r["cin"], r["cout"], r["kvol"], r["split"])
        g = groups.setdefault(key, {"ms": 0.0, "n": 0, "by": 0, "fl": 0, "extra": 0, "miss": 0, "n_out": 0})
        g["ms"] += r["ms"]
        g["n"] += 1
        g["n_out"] += r["n_out"]
        cand = pairs_of.get((r["cin"], r["cout"], r["kvol"], r["n_out"]))
        if not cand:
            g["miss"] += 1
            continue
        # layers of one stage share the rulebook (same n_out -> same R); distinct tables of equal n_out average
        R = sum(cand) // len(cand)
        by, fl, extra = conv_algorithmic(r, R)
        g["by"] += by
        g["fl"] += fl
        g["extra"] += extra
    if not groups:
        return None, {}
    if want is None:
        key = max(groups, key=lambda k: groups[k]["ms"])
    elif want in groups:
        key = want
    else:
        return None, {}
    cin, cout, K, split = key
    g = groups[key]
    sec = g["ms"] * 1e-3
    counted = max(g["n"] - g["miss"], 1)
    by, fl, extra = (g[k] * g["n"] // counted for k in ("by", "fl", "extra"))
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
    n_out_avg = g["n_out"] // g["n"]
    kname = trace_kernel_name(cin, cout, K, split, n_out_avg)
    s = 2 if split == 2 else 4
    roof.update({"traffic": None, "kernel": "%s K=%d" % (kname, K),
                 "precision": ("bf16 rows and weights, fp32 accumulate" if split == 2 else
                               "fp16 hi/lo operand pairs, 3 MFMA products, fp32 accumulate (fp32-grade)" if split else "fp32 MFMA"),
                 "launches": g["n"], "avg_launch_us": round(g["ms"] * 1e3 / g["n"], 2),
                 "algorithmic_flops_per_launch": fl // g["n"], "algorithmic_bytes_per_launch": by // g["n"],
                 "extra_written_bytes_per_launch": extra // g["n"],
                 "arithmetic_intensity_flop_per_byte": round(ai, 1)})
    if K == 27:
        # SubM layer (n_in = n_out): every input row once, every output row once, the neighbour table, the filter bank
        roof["compulsory_bytes"] = n_out_avg * cin * s + n_out_avg * cout * s + 4 * K * n_out_avg + K * cin * cout * s
    # HBM traffic of that kernel comes from separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; PMC cannot be
    # read from inside the process).  The committed summary of the last such run is attached when it belongs to the
    # same kernel; otherwise null.
    try:
        pm = json.load(open(PMC_SUMMARY))
        if pm.get("kernel_key") != [cin, cout, K, split]:      # the summary's main entry is another kernel: its table of the
            cal = pm.get("fetch_calibration")                   # other K = 27 kernels of the same passes may hold this one
            pm = next((v for v in pm.get("other_k27_kernels", {}).values() if v.get("kernel_key") == [cin, cout, K, split]), {})
            if pm and cal and "fetch_calibration" not in pm:
                pm["fetch_calibration"] = cal
        if pm.get("kernel_key") == [cin, cout, K, split]:
            roof["traffic"] = pm["traffic_bytes_per_launch"]
            roof["traffic_source"] = "profiles/%s (rocprofv3 --pmc, separate passes, same command)" % os.path.basename(PMC_SUMMARY)
            # what the HBM interface really moved during this kernel, against the 8 TB/s peak
            roof["hbm_measured_frac"] = round(pm["traffic_bytes_per_launch"] / (g["ms"] * 1e-3 / g["n"]) / (PEAK_HBM_GBS * 1e9), 4)
            if "fetch_calibration" in pm:
                roof["fetch_calibration"] = pm["fetch_calibration"]
    except Exception:
        pass
    per_kernel = {"%dx%d_k%d%s" % (k[0], k[1], k[2], "_split" if k[3] else ""):
                  {"ms_total": round(v["ms"], 3), "launches": v["n"]} for k, v in groups.items()}
    return roof, per_kernel



# ------------------------------------------------------------------------------------------------ per-kernel rooflines
def api_probe(wl, stage):
    """One untimed step with HIP events around every C-ABI call (dualfusion/apitimer.py), on the per-module path
    (DF3D_EXECUTOR=0: the same kernels as the native executor launches, but the rulebook entries are visible one by one)."""
    from dualfusion.apitimer import ApiTimer, summarize
    old = os.environ.get("DF3D_EXECUTOR")
    os.environ["DF3D_EXECUTOR"] = "0"
    try:
        wl.step(0, stage)                                     # the module path's own caches
        torch.cuda.synchronize()
        t = ApiTimer().start()
        try:
            wl.step(0, stage)
        finally:
            recs = t.stop()
    finally:
        if old is None:
            os.environ.pop("DF3D_EXECUTOR", None)
        else:
            os.environ["DF3D_EXECUTOR"] = old
    return summarize(recs)


def roofline_by_kernel(probe, meta_timer, api, min_us=30.0):
    """Every kernel (class) of one step that takes >= `min_us`, with its own roofline: SURVEY.md section 8(d) algorithmic bytes
    (or flops) / measured time against the roof that bounds it.  Convolutions: HIP events inside the library (probe step of
    the timed configuration) by (cin, cout, K); everything else: HIP events around the C-ABI entry point (api_probe)."""
    out = []
    keys = sorted({(r["cin"], r["cout"], r["kvol"], r["split"]) for r in probe.records})
    first_rows = None
    for k in keys:
        ro, _ = roofline_from_timer(probe, meta_timer, want=k)
        if ro is None:
            continue
        us = ro["avg_launch_us"] * ro["launches"]
        if k[0] <= 5 and k[2] == 27:
            first_rows = [r["n_out"] for r in probe.records if (r["cin"], r["kvol"]) == (k[0], 27)][0]
        if us < min_us:
            continue
        out.append({"kernel": "conv %d->%d K=%d" % k[:3], "us_per_step": round(us, 1), "launches": ro["launches"],
                    "bound": ro["bound"], "achieved": ro["achieved"], "peak": ro["peak"], "unit": ro["unit"], "frac": ro["frac"],
                    "what": "%s; %s" % (ro["kernel"], ro["precision"])})
    # rulebook bytes: 16 N_in + 8 R + 16 N_out per distinct table of the step (pairs from the metadata pass)
    tables = {(r["kvol"], r["n_out"]): r["pairs"] for r in meta_timer.records if r["kvol"] == 27}
    rb_bytes = sum(32 * n + 8 * R for (_, n), R in tables.items())
    for name, e in api.items():
        us = e["ms"] * 1e3
        by, fl, bound = e["bytes"], e["flops"], e["bound"]
        if name == "rulebook":
            by, fl = rb_bytes, 0
        if name.startswith("df3d_hard_voxelize") and first_rows:
            by += 36 * first_rows
        if us < min_us:
            continue
        ent = {"kernel": name, "us_per_step": round(us, 1), "launches": e["calls"], "what": e["what"]}
        if bound is not None and not e["unknown"] and us > 0:
            peak_tf = PEAK_BF16_MFMA_TF / 3.0
            if bound == "mfma" or (fl and by and fl / by >= peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)):
                ach = fl / (us * 1e-6) / 1e12
                ent.update({"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak_tf, 1), "unit": "TFLOP/s",
                            "frac": round(ach / peak_tf, 4)})
            else:
                ach = by / (us * 1e-6) / 1e9
                ent.update({"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                            "frac": round(ach / PEAK_HBM_GBS, 4)})
        out.append(ent)
    out.sort(key=lambda d: -d["us_per_step"])
    return out

# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(wl, n_sweeps=20):
    """The reference's CPU path on this box's host cores, same workload, same weights, one sweep at a time:
      voxelize        oracle/_ref/voxel_layer.so `hard_voxelize` (TF/mmdet3d/ops/voxel/src/voxelization_cpu.cpp)
      sparse backbone oracle/_ref/sparse_conv_ext.so `get_indice_pairs_3d` + `indice_conv_fp32` (the reference's own
                      compiled CPU rulebook + gather/GEMM/scatter, spconv_ops.h:27-141,260-361), BN / ReLU in numpy
      camera fusion   the oracle port (tests/oracle_models.py: projection, gate, ACTR through torch-CPU fp32) -- the
                      reference has no CPU build of its MSDA op
      neck+head+loss  the mirror modules' torch composition on the CPU (= the reference's own torch layers)
    Thread settings and sample: see the comment above the timing loop."""
    import oracle_models as om
    from oracle import oracle as orc
    from oracle import ref
    from dualfusion import synth
    if not (ref.available("sparse_conv_ext") and ref.available("voxel_layer")):
        return {"value": None, "unit": "sweeps/s", "cores": 0, "kind": "reference",
                "sample": "oracle/_ref/*.so missing on this box (built where /root/reference exists)"}
    model = wl.model
    sd_all = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    sd = {k[len("hot_path.backbone."):]: v for k, v in sd_all.items() if k.startswith("hot_path.backbone.")}
    sd_f = {k[len("hot_path.fusion."):]: v for k, v in sd_all.items() if k.startswith("hot_path.fusion.")}
    import copy
    neck_cpu = copy.deepcopy(model.neck).cpu().eval()
    head_cpu = copy.deepcopy(model.bbox_head).cpu().eval()
    all_threads = torch.get_num_threads()
    fusion_on = wl.name == "cp_fusion"

    def one_sweep(seed, fr):
        st = {}
        pts = synth.nusc_sweep(seed=seed)
        t0 = time.perf_counter()
        ov, oc, on = ref.hard_voxelize(pts, synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, 120000)
        feats = orc.mean_vfe(ov, on)
        st["voxelize"] = time.perf_counter() - t0
        coors = np.concatenate([np.zeros((len(oc), 1), np.int32), oc], 1)
        fuse = None
        if fusion_on:
            from dualfusion.fusion import CP_DEPTH_THRES
            cam = fr["cam"]
            img = {n: cam["feats"][:, i].cpu().numpy() for i, n in enumerate(synth.NUSC_CAMS)}
            calib = {n: (cam["calib"]['lidar2cam_' + n.lower().lstrip('cam_')].cpu().numpy(),
                         cam["calib"]['cam_intrinsic_' + n.lower().lstrip('cam_')].cpu().numpy()) for n in synth.NUSC_CAMS}
            hw = tuple(cam["shape"][:2])

            def fuse(c2, c3, c4):
                t1 = time.perf_counter()
                # rows in spconv's GPU order (sorted by flat index), which the adapter's "last writer wins" assumes
                lv = [tuple(np.ascontiguousarray(a) for a in om.sort_rows(c.indices, c.features)) for c in (c2, c3, c4)]
                with om.using(orc):
                    out = om.centerpoint_fusion(sd_f, lv, img, calib, hw, synth.NUSC_CAMS, synth.NUSC_VOXEL,
                                                synth.NUSC_RANGE, 2.0 / 3.0, CP_DEPTH_THRES)
                st["fusion_port"] = time.perf_counter() - t1
                c4.indices, c4.features, c4.rulebooks = lv[2][0], out, {}
                return c4
        om.STAGE_SECONDS.clear()
        t0 = time.perf_counter()
        with om.using(ref):
            bev, _ = om.centerpoint_backbone(sd, feats, coors, 1, [1440, 1440, 40], fuse=fuse)
        st["backbone"] = time.perf_counter() - t0 - st.get("fusion_port", 0.0)
        st["backbone_rulebook"] = om.STAGE_SECONDS.get("rulebook", 0.0)
        st["backbone_conv"] = om.STAGE_SECONDS.get("conv", 0.0)
        t0 = time.perf_counter()
        with torch.no_grad():
            x = neck_cpu.forward_reference(torch.from_numpy(np.ascontiguousarray(bev)))
            preds = head_cpu.forward_reference(x)
            ex = {k: [a[:1].cpu() for a in v] for k, v in fr["targets"].items()}
            head_cpu.loss(ex, preds, {})
        st["neck_head_loss"] = time.perf_counter() - t0
        st["total"] = sum(v for k, v in st.items() if k in ("voxelize", "backbone", "fusion_port", "neck_head_loss"))
        return st

    # The reference's CPU algorithm is single-threaded C++ around torch::mm (spconv_ops.h:260-361): whether BLAS threads
    # help depends on the box (on a 128-core EPYC the small per-offset GEMMs get SLOWER with all threads).  One probe
    # sweep per thread setting {1, 16, all} (the very first sweep is the warm-up and is not counted), then `n_sweeps`
    # timed sweeps at the fastest setting; `value` = 1 / median of those.
    t_all = time.perf_counter()
    # (all threads were probed in rounds 2-3: 15-20 s per sweep on the 128-core box against 4 s at 16 -- not probed any more)
    settings = sorted(set([1, min(16, all_threads)]))
    torch.set_num_threads(min(16, all_threads))
    one_sweep(0, wl.frames[0])                                   # warm-up (page-in, allocator, lazy inits)
    probe = {}
    for threads in settings:
        torch.set_num_threads(threads)
        probe[threads] = one_sweep(1, wl.frames[1 % len(wl.frames)])
    best = min(settings, key=lambda t: probe[t]["total"])
    torch.set_num_threads(best)
    runs = [one_sweep(2 + i, wl.frames[(2 + i) % len(wl.frames)]) for i in range(max(1, n_sweeps))]
    torch.set_num_threads(all_threads)
    wall = time.perf_counter() - t_all
    stages = ("voxelize", "backbone", "backbone_rulebook", "backbone_conv", "fusion_port", "neck_head_loss", "total")
    med = {k: round(float(np.median([r.get(k, 0.0) for r in runs])), 4) for k in stages}
    return {"value": round(1.0 / med["total"], 4), "unit": "sweeps/s", "cores": int(best), "kind": "reference",
            "cpu_model": cpu_model_string(), "host_cpu_count": os.cpu_count() or 0,
            "seconds_per_sweep_median": med,
            "probe_seconds_per_sweep_by_threads": {str(t): {k: round(v.get(k, 0.0), 4) for k in stages}
                                                   for t, v in probe.items()},
            "sample": "3 untimed sweeps (1 warm-up, 1 probe at each of %s threads), then %d whole synthetic sweeps at the faster "
                      "setting (%d threads; median reported); voxelize and the sparse backbone on the reference's compiled "
                      "CPU ops (oracle/_ref: hard_voxelize, get_indice_pairs_3d, indice_conv_fp32)%s, neck + head + loss "
                      "on torch CPU fp32; %.0f s wall" % (
                          settings, len(runs), best,
                          ", camera fusion through the oracle port (kind 'port' for that stage)" if fusion_on else "", wall)}


# ------------------------------------------------------------------------------------------------ the other configs
SIDE_CONFIGS = (("cp_lidar", "configs[0] shape", "split", "detect"), ("tf_fusion", "configs[2]", "bf16", "detect"),
                ("vr_fusion", "configs[4]", "split", "detect"),
                # (VERDICT r4 "next" 5: the training step where the driver's record sees it)
                ("cp_fusion", "configs[1], training step: forward + losses + backward + optimizer", "split", "train"),
                # (VERDICT r5 "next" 1: configs[3]'s per-rank body -- TransFusion-L + 3D-DF bs 4 as a training step)
                ("tf_fusion", "configs[3] per rank, training step: forward + Hungarian losses + backward + grad clip + AdamW", "bf16",
                 "train"))


def side_configs(args, steps=12):
    """BASELINE configs[0] (shape), [2] and [4] after the headline, ~`steps` timed steps each with the same protocol (every
    distinct frame once, warm-up, barrier, K steps, barrier): compact numbers for the END of the JSON line, where the
    driver's record keeps them.  Each runs as `python bench.py --workload ...` in a process of its own, after this process
    has released its device memory (in-process they ran 10-50 % slower than alone: allocator state and streams of the headline
    workload).  A failure of one of them is reported in its entry and does not fail the bench line."""
    import subprocess
    out = {}
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    for name, cfg, prec, stage in SIDE_CONFIGS:
        if name == args.workload and stage == "detect":
            continue
        key = name if stage == "detect" else "%s_%s" % (name, stage)
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", str(steps), "--warmup", "4",
               "--frames", "4", "--conv-precision", prec, "--no-cpu-baseline", "--no-extra-passes", "--no-kernel-timing",
               "--no-side-configs", "--inflight", "1", "--stage", stage]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                raise RuntimeError("rc %d: %s" % (r.returncode, r.stderr.strip().splitlines()[-1][:100] if r.stderr.strip() else ""))
            d = json.loads(line[-1])
            out[key] = {"cfg": cfg, "ms_per_step": d["ms_per_step"], "bs": d["config"]["sweeps_per_gpu_per_step"],
                         "value": d["value"], "unit": d["unit"],
                         "dtype": {"split": "f32(fp16 hi+lo operands, 3 products)", "bf16": "bf16", "fp32": "f32"}[prec], "steps": steps}
        except Exception as ex:                                  # noqa: BLE001
            out[key] = {"cfg": cfg, "error": repr(ex)[:120]}
    return out


# ------------------------------------------------------------------------------------------------ main
def precision_probe(wl, ops, frames=2):
    """Evidence for `dtype` on the bench line itself: the dense BEV map (after voxelisation, the 21 sparse convolutions and the
    camera fusion: every matrix-core GEMM of the hot path has contributed) and the detection head's output maps (after the
    neck and the head's convolutions) of the first frames in the headline mode ("split": fp16 hi + lo operands), in "split3"
    (three bf16 parts) and with every convolution on the exact-fp32 MFMA kernels, each against the exact-fp32 result:
    max |difference| / max |reference|."""
    import torch
    old = ops.CONV_PRECISION
    res = {"what": "max |x - x_fp32| / max |x_fp32| over the dense BEV map [B, 256, 180, 180] and over the detection head's "
                   "output maps of %d frame(s), x = a precision mode's result, x_fp32 = the ALL-fp32 result (round 6, "
                   "ops.reference_arithmetic): every convolution on v_mfma_f32_16x16x4_f32 (exact fp32 products) AND the "
                   "feed-forward blocks / query linears / image projection / value GEMM on the library's fp32 GEMMs; "
                   "'fp32_convs' = the exact-fp32 convolutions with the other GEMMs on the fp16 hi + lo kernels (round 5's "
                   "yardstick)" % frames}
    bev, heads = {}, {}
    try:
        for mode in ("fp32", "fp32_convs", "split", "split3"):
            ops.CONV_PRECISION = "fp32" if mode == "fp32_convs" else mode
            ops.ALL_FP32 = mode == "fp32"
            bev[mode], heads[mode] = [], []
            for k in range(frames):
                fr = wl.frames[k % len(wl.frames)]
                bd, example = wl.fresh_inputs(fr)
                bev[mode].append(wl.step(k, "hot_path").float().clone())
                wl._staged = {}
                with torch.no_grad():
                    x, _ = wl.model.hot_path(fr["points"], batch_dict=bd, example=example)
                    preds = wl.model.bbox_head(x)
                heads[mode].append(torch.cat([v.float().reshape(-1) for d in preds for _, v in sorted(d.items())]).clone())
    finally:
        ops.CONV_PRECISION, ops.ALL_FP32 = old, False
    torch.cuda.synchronize()
    for mode in ("split", "split3", "fp32_convs"):
        worst, hworst = 0.0, 0.0
        for a, b, ha, hb in zip(bev[mode], bev["fp32"], heads[mode], heads["fp32"]):
            worst = max(worst, float((a - b).abs().max() / b.abs().max()))
            hworst = max(hworst, float((ha - hb).abs().max() / hb.abs().max()))
        res[mode] = {"bev_map": float("%.3e" % worst), "head_maps": float("%.3e" % hworst)}
    res["bev_map_scale"] = float("%.4g" % float(bev["fp32"][0].abs().max()))
    res["head_maps_scale"] = float("%.4g" % float(heads["fp32"][0].abs().max()))
    return res


def timed_steps(wl, stage, steps, first, barrier, reduce_losses):
    barrier()
    # no cyclic-GC pause inside the timed steps (a generation-2 pass over a process holding ~10^5 tensor wrappers takes
    # milliseconds: more than the queueing thread's slack per frame); reference counting frees everything the steps create
    import gc
    gc_was = gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        t0 = time.perf_counter()
        out = None
        for k in range(steps):
            out = wl.step(first + k, stage)
            if isinstance(out, dict) and stage in ("detect", "train"):
                out = reduce_losses(out)
        barrier()                                  # synchronises the device (all streams) and the ranks
        el = time.perf_counter() - t0
    finally:
        if gc_was:
            gc.enable()
    return el, out


def host_blocked_s(wl):
    """Host seconds a workload's detector has spent WAITING so far (for its helper thread's results, for a frame slot the GPU
    has not released yet): back-pressure, not queueing work."""
    ahead = getattr(getattr(getattr(wl, "model", None), "hot_path", None), "_ahead", None)
    st = getattr(ahead, "stats", None)
    return (st["take_wait_s"] + st["slot_wait_s"]) if st else 0.0


def timed_steps_alternating(wls, streams, stage, steps, first, barrier):
    """The same K steps with len(wls) frames in flight from ONE host thread: step k is queued on stream k % F by detector
    replica k % F (frames are independent).  Every count round trip of a frame is taken by its replica's helper thread a frame
    ahead (dualfusion/prefetch.py), so the queueing thread never waits for the device and the kernels of F frames overlap on
    the GPU.  Returns (elapsed, per-frame latencies in ms: queue start -> last kernel done, from events)."""
    F = len(wls)
    marks = []
    barrier()
    blocked0 = sum(host_blocked_s(w) for w in wls)
    t0 = time.perf_counter()
    for k in range(steps):
        st = streams[k % F]
        with torch.cuda.stream(st):
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(st)
            wls[k % F].step(first + k, stage)
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(st)
        marks.append((e0, e1))
    queued = time.perf_counter() - t0                    # the host has queued every step ...
    queued -= sum(host_blocked_s(w) for w in wls) - blocked0     # ... minus the time it waited for the GPU / the helper thread
    barrier()
    el = time.perf_counter() - t0
    return el, [a.elapsed_time(b) for a, b in marks], queued


def timed_steps_in_flight(wls, stage, steps, first, barrier):
    """The same K steps with len(wls) frames in flight: one host thread + HIP stream + detector replica each, step k goes to
    replica k % F (frames are independent; the host syncs of a step -- voxel count, three rulebook sizes, longest camera
    list -- then stall one thread while the other keeps the GPU fed).  Single rank only: collectives from two threads of one
    process would interleave differently on different ranks."""
    import threading
    F = len(wls)
    streams = [torch.cuda.Stream() for _ in wls]
    errors = []

    def work(t):
        try:
            with torch.cuda.stream(streams[t]):
                for k in range(t, steps, F):
                    wls[t].step(first + k, stage)
        except Exception as e:             # noqa: BLE001  (re-raised on the main thread)
            errors.append(e)
    barrier()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(t,)) for t in range(F)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    barrier()
    if errors:
        raise errors[0]
    return time.perf_counter() - t0


def note(msg):
    if os.environ.get("DF3D_BENCH_TRACE"):
        print("bench.py: " + msg, file=sys.stderr, flush=True)


def main():
    args = parse()
    from dualfusion import dist as D
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(D.launch_ranks(args.gpus, os.path.abspath(__file__), sys.argv[1:]))
    protocol = args.workload == "protocol"
    if not protocol:
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback on the product path)"
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available() and args.backend != "gloo"
    if use_gpu and os.environ.get("DF3D_BENCH_NICE", "1") == "1":
        # The queueing thread has ~1 ms of slack per 2.7 ms frame; on a shared host a descheduled interpreter eats it (one run
        # in ~10 on the pool's boxes reads 3.0 instead of 2.65 ms per step with identical kernel times).  A production
        # serving / training process would be pinned and prioritised the same way; needs CAP_SYS_NICE, silently skipped without.
        try:
            os.nice(-10)
        except OSError:
            pass
    if use_gpu:
        torch.cuda.set_device(local)
    dev = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    # nccl == RCCL on ROCm.  At N = 1 a one-rank group is created as well, so the step's collectives (reduce of the loss
    # scalars, gradient buckets) go through RCCL exactly as at N > 1 -- `collective_backend` on the JSON line says so.
    # (The reference's reduce_dict returns its input at one rank, utils.py:164-166; taking that shortcut was measured in
    # round 4 and is not faster: 2.755 against 2.728 ms per step.)
    try:
        rank, local, world = D.init_from_env(args.backend or ("nccl" if use_gpu else "gloo"), single=use_gpu and not protocol)
    except Exception as e:                                       # noqa: BLE001
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            raise
        print("bench.py: one-rank process group not available (%s); running without" % (e,), file=sys.stderr)
        rank, local, world = 0, 0, 1
    assert world == max(1, args.gpus), "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)
    precision = args.conv_precision or ("bf16" if args.workload == "tf_fusion" else "split")
    if not protocol:
        from dualfusion import ops
        ops.CONV_PRECISION = precision
    wl = make_workload(args, rank, world, dev)
    stage = args.stage

    def barrier():
        D.barrier(dev if use_gpu else None)

    def reduce_losses(losses):
        # CP/det3d/torchie/trainer/utils.py:157-183: rank 0 holds the average.  Only a dict of loss tensors is reduced: the
        # Voxel-RCNN tree's step ends at the backbone and hands back its batch_dict (no detection head in the reference path)
        if not all(hasattr(v, "reshape") for v in losses.values()):
            return losses
        return D.reduce_dict(losses)

    # Setup, before the W warm-up steps: every distinct frame once.  The timed steps rotate through the frames, and the
    # first visit of a frame sizes the caching allocator's blocks for ITS voxel counts and fills the address-keyed tables;
    # with W < frames that would land in the timed region (3.75 vs 3.3 ms per step at W = 2).
    for k in range(len(getattr(wl, "frames", ()))):
        out = wl.step(k, stage)
        if isinstance(out, dict) and stage in ("detect", "train"):
            out = reduce_losses(out)
    for k in range(args.warmup):
        out = wl.step(k, stage)
        if isinstance(out, dict) and stage in ("detect", "train"):
            out = reduce_losses(out)
    barrier()
    # Roofline: HIP events around EVERY conv launch cost the timed region ~6 % (30 launches per frame).  The roofline
    # needs the dominant kernel only: one untimed probe step with all events finds it (and yields the per-kernel
    # table), the timed region then records events around its launches alone.
    timer = probe = meta_timer = None
    first_timed = args.warmup
    kernel_timing = not (args.no_kernel_timing or protocol)
    if kernel_timing:
        probe = ops.KernelTimer()
        probe.start()
        wl.step(args.warmup, stage)            # the next step of the sequence (its head was prefetched by the last warm-up step)
        torch.cuda.synchronize()
        probe.stop()
        first_timed = args.warmup + 1
        tot = {}
        for r in probe.records:
            if r["kvol"] == 27:                  # the 3-D sparse convolutions (the neck / head reuse the kernel with K = 9)
                k = (r["cin"], r["cout"], r["kvol"])
                tot[k] = tot.get(k, 0.0) + r["ms"]
        # (a SAMPLE of its launches: every 5th -- its four launches per frame take turns; events around all of them cost the
        # timed region ~5 %, 3.00 against 2.85 ms per step)
        timer = ops.KernelTimer(only=max(tot, key=tot.get), every=5) if tot else ops.KernelTimer(every=5)
        timer.start()
    note("timed region")
    elapsed, out = timed_steps(wl, stage, args.steps, first_timed, barrier, reduce_losses)
    if timer is not None:
        timer.stop()
    elapsed = D.max_over_ranks(elapsed, dev)
    wl.check(out, stage)
    extra = {}
    if stage == "train":
        assert args.workload in ("cp_lidar", "cp_fusion", "tf_fusion"), "--stage train: the CenterPoint / TransFusion detectors"
    if not (args.no_extra_passes or protocol or stage == "train"):
        # the same K steps ending at the dense BEV tensor (round 1's step), and both stages on the exact-fp32 kernels
        skip = set(x for x in args.skip_passes.split(",") if x)
        if stage == "detect" and "hot_path" not in skip:
            note("hot_path pass")
            wl.step(0, "hot_path")
            e, o = timed_steps(wl, "hot_path", args.steps, args.warmup, barrier, reduce_losses)
            wl.check(o, "hot_path")
            extra["hot_path"] = D.max_over_ranks(e, dev)
        if (stage == "detect" and world == 1 and args.workload in ("cp_fusion", "cp_lidar") and args.inflight > 1
                and "in_flight" not in skip):
            F = args.inflight
            note("in_flight pass")
            wls = [wl] + [make_workload(args, rank, world, dev) for _ in range(F - 1)]
            if args.prefetch:
                streams = [torch.cuda.Stream() for _ in wls]
                for w in wls:
                    w.stride = F
                nf = len(wl.frames)
                timed_steps_alternating(wls, streams, stage, 2 * nf, 0, barrier)        # every replica sees every frame once
                if_steps = max(args.steps, 48)                    # (the pipeline fills and drains once per pass)
                e, lat, queued = timed_steps_alternating(wls, streams, stage, if_steps, 2 * nf, barrier)
                extra["in_flight"], extra["in_flight_latency_ms"] = e * args.steps / if_steps, lat
                extra["in_flight_host_ms"] = queued / if_steps * 1e3
                for w in wls:
                    w.stride = 1
            else:
                timed_steps_in_flight(wls, stage, 2 * F, 0, barrier)           # warm the replicas
                extra["in_flight"] = timed_steps_in_flight(wls, stage, args.steps, args.warmup, barrier)
            for w in wls[1:]:
                w.close()
            del wls
        if precision == "split" and args.workload in ("cp_fusion", "cp_lidar"):
            # the like-for-like companions of the headline: every convolution fp32-grade -- "split3" (three bf16 parts per
            # operand, six products, ~1e-7) and "fp32" (the exact-fp32 MFMA kernels)
            for mode in ("split3", "fp32"):
                if mode in skip:
                    continue
                ops.CONV_PRECISION = mode
                note(mode + " passes")
                try:
                    for st in (["detect", "hot_path"] if stage == "detect" else ["hot_path"]):
                        # every distinct frame once in this mode before its timed steps, as in the set-up of the headline
                        # pass: a frame's first visit in a mode with wider rows sizes new allocator blocks (hipMalloc + a
                        # device synchronisation each) -- with three warm-up steps five of the eight frames paid that
                        # inside the timed steps (4.6 ms read 7.7-10.8 ms when the allocator was already fragmented)
                        nwarm = max(3, len(getattr(wl, "frames", ())))
                        for k in range(args.warmup - nwarm, args.warmup):      # ends where the timed steps begin (prefetch)
                            o = wl.step(k, st)
                            if isinstance(o, dict) and st == "detect":
                                reduce_losses(o)
                        e, o = timed_steps(wl, st, args.steps, args.warmup, barrier, reduce_losses)
                        wl.check(o, st)
                        extra[mode + "_" + st] = D.max_over_ranks(e, dev)
                finally:
                    ops.CONV_PRECISION = precision
    if kernel_timing:
        # metadata pass, outside the timed region: every frame once more, counting the valid rulebook pairs of every
        # conv launch (the unit the algorithmic bytes are stated in)
        note("metadata pass")
        meta_timer = ops.KernelTimer(count_pairs=True)
        meta_timer.start()
        for k in range(len(getattr(wl, "frames", [0]))):
            wl.step(k, stage)
        torch.cuda.synchronize()
        meta_timer.stop()
    precision_evidence = None
    if (rank == 0 and world == 1 and stage == "detect" and precision == "split" and args.workload in ("cp_fusion", "cp_lidar")
            and not args.no_extra_passes and not protocol):
        try:
            note("precision probe")
            precision_evidence = precision_probe(wl, ops)
        except Exception as e:                                   # noqa: BLE001  (a measurement aid must not fail the bench line)
            print("bench.py: precision probe failed: %r" % (e,), file=sys.stderr)
    if use_gpu and not protocol:
        # the fp16 operand split is range-limited (csrc/common.h): a value it could not hold raises a device flag instead of
        # passing silently -- a bench line must not be printed over such a run
        ops.check_split_overflow()
    api = None
    if kernel_timing and rank == 0 and stage in ("detect", "hot_path"):
        try:
            note("api probe")
            api = api_probe(wl, stage)
        except Exception as e:                                   # noqa: BLE001  (a measurement aid must not fail the bench line)
            print("bench.py: api probe failed: %r" % (e,), file=sys.stderr)
    if rank == 0:
        units = args.steps * wl.batch * world
        per_step = lambda e: round(e / args.steps * 1e3, 4)      # noqa: E731
        res = {
            "metric": getattr(wl, "metric", "nuScenes sweeps/sec (0.075 m voxel, ~60k pts)"), "value": round(units / elapsed, 3),
            "unit": wl.unit_name + "/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": per_step(elapsed), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"split": "f32 (fp32-grade throughout: every matrix-core GEMM -- C>=32 sparse convs, neck / head convs, FFN, "
                               "query linears, image projection, value GEMM -- takes its fp32 operands as fp16 hi+lo pairs, "
                               "22 significand bits, 3 MFMA products, fp32 accumulate: <= 4e-6 of scale against float64, the "
                               "grade of the exact-fp32 MFMA kernels; everything else exact fp32)",
                      "fp32": "f32 (exact fp32 MFMA convolutions; FFN / query linears / image side on fp16 hi+lo operands)",
                      "split3": "f32 (C>=32 convolutions: operands in three bf16 parts, 6 MFMA products, fp32 accumulate, "
                                "fp32-grade with fp32's exponent range; FFN / query linears / image side on fp16 hi+lo operands)",
                      "bf16": "bf16 sparse convs (bf16 rows and weights, fp32 accumulate and epilogue); fusion adapter, "
                              "ACTR and C<=16 layers f32"}[precision] if not protocol else "none",
            "data": "synthetic",
            "config": {"workload": wl.describe(), "stage": stage, "sweeps_per_gpu_per_step": wl.batch,
                       "distinct_frames_per_rank": len(getattr(wl, "frames", [0])),
                       "setup": "one untimed pass over the distinct frames before the W warm-up steps (allocator block sizes "
                                "and address-keyed tables of every frame exist before the timed region, whatever W is)",
                       "global_batch": wl.batch * world,
                       "parallelism": "dp%d (frames sharded over the ranks; the only collective is the reduce of the loss "
                                      "scalars%s)" % (world, ", RCCL" if use_gpu and world > 1 else "")},
            "collective_backend": (torch.distributed.get_backend() if D.is_dist() else None), "world_size": world,
        }
        if stage == "train" and args.workload == "tf_fusion":
            res["config"]["training"] = {"trainable_parameters": getattr(wl, "n_params", None),
                                         "optimizer": "AdamW (fused) lr 1e-4 wd 0.01, grad_clip max_norm 0.1 (TF/configs/"
                                                      "transfusion_nusc_voxel_F.py:302-303)",
                                         "loss_logging": "device scalars (loss terms, grad_norm) reduced over the ranks; no host copy "
                                                         "inside the step",
                                         "gradient_reduction": "GradBucketReducer: %d bucket(s) of <= 16 MB, all-reduces launched in "
                                                               "bucket order from post-accumulate-grad hooks during backward" % len(wl.reducer.buckets)}
        elif stage == "train":
            async_log = os.environ.get("DF3D_TRAIN_ASYNC_LOG", "1") == "1"
            res["config"]["training"] = {"trainable_parameters": getattr(wl, "n_params", None), "optimizer": "AdamW (fused)",
                                         "loss_logging": ("async: hm_loss / loc_loss_elem go to pinned host memory without a host "
                                                          "wait (a trainer reads them every N steps behind an event); NOT the "
                                                          "reference's per-step .cpu() of parse_second_losses -- "
                                                          "DF3D_TRAIN_ASYNC_LOG=0 times that" if async_log else
                                                          "sync: the reference's per-step .cpu() copies of the logged losses "
                                                          "(trainer.py parse_second_losses) inside the timed step"),
                                         "gradient_reduction": "GradBucketReducer: %d bucket(s) of <= 16 MB, all-reduces launched in "
                                                               "bucket order from post-accumulate-grad hooks during backward" % len(wl.reducer.buckets)}
        if stage in ("detect", "train") and isinstance(out, dict) and "encoded_spconv_tensor" not in out:
            res["reduced_losses"] = {k: [round(float(x), 5) for x in v.reshape(-1).float().cpu()] for k, v in out.items()
                                     if ("loss" in k and not k.endswith("_elem")) or k == "matched_ious"}
        if "in_flight" in extra:
            res["in_flight"] = {"frames_in_flight": args.inflight, "ms_per_step": per_step(extra["in_flight"]),
                                "value": round(units / extra["in_flight"], 3), "unit": res["unit"],
                                "what": "the same K detector steps in ONE process, step k queued on stream k %% %d by detector "
                                        "replica k %% %d from one host thread; the count round trips of a frame are taken a "
                                        "frame ahead by the replica's helper thread; never `value`" % (args.inflight, args.inflight)}
            if "in_flight_host_ms" in extra:
                # (time spent waiting for the helper thread / for a frame slot the GPU still holds is not counted)
                res["in_flight"]["host_ms_per_frame"] = round(extra["in_flight_host_ms"], 3)
            lat = extra.get("in_flight_latency_ms")
            if lat:
                res["in_flight"]["frame_latency_ms"] = {"median": round(float(np.median(lat)), 3), "max": round(max(lat), 3),
                                                         "what": "per frame, events on its stream: first kernel queued -> last done"}
        if "hot_path" in extra:
            res["hot_path"] = {"ms_per_step": per_step(extra["hot_path"]), "value": round(units / extra["hot_path"], 3),
                               "unit": wl.unit_name + "/s",
                               "what": getattr(wl, "hot_path_what", "the same K steps ending at the dense BEV tensor "
                                       "[B,256,180,180] (no neck / head / losses / reduce): round 1's step")}
        if "fp32_detect" in extra or "fp32_hot_path" in extra:
            res["fp32"] = {"what": "every convolution (sparse backbone, BEV neck, detection head) on the exact-fp32 MFMA "
                                   "kernels (--conv-precision fp32)"}
            if "fp32_detect" in extra:
                res["fp32"]["ms_per_step"] = per_step(extra["fp32_detect"])
                res["ms_per_step_fp32"] = per_step(extra["fp32_detect"])
            if "fp32_hot_path" in extra:
                res["fp32"]["hot_path_ms_per_step"] = per_step(extra["fp32_hot_path"])
        if "split3_detect" in extra or "split3_hot_path" in extra:
            res["split3"] = {"what": "every C >= 32 convolution (sparse backbone, BEV neck, detection head) with operands in THREE "
                                     "bf16 parts (hi + mid + lo = the fp32 value exactly) and six matrix-core products, fp32 "
                                     "accumulate: fp32-grade (<= 4e-6 of scale against float64, like the exact-fp32 kernels) "
                                     "(--conv-precision split3)"}
            if "split3_detect" in extra:
                res["split3"]["ms_per_step"] = per_step(extra["split3_detect"])
                res["ms_per_step_split3"] = per_step(extra["split3_detect"])
            if "split3_hot_path" in extra:
                res["split3"]["hot_path_ms_per_step"] = per_step(extra["split3_hot_path"])
        if kernel_timing:
            roof, _ = roofline_from_timer(timer, meta_timer)
            _, per_kernel = roofline_from_timer(probe, meta_timer)        # all conv kernels, from the untimed probe step
            if roof is not None:
                roof["measured_over"] = ("the timed region (HIP events around every 5th launch of this kernel only -- it was "
                                         "picked by an untimed probe step with events on every launch)")
            res["roofline"] = roof
            res["conv_kernel_ms_probe_step"] = per_kernel
            # the other 3-D sparse-conv kernels against the same roofs, from the launches of the untimed probe step (the
            # timed region carries events around the dominant kernel only); round 1 / 2 quoted conv4 = 128x128 K=27
            others = {}
            for k in sorted({(r["cin"], r["cout"], r["kvol"], r["split"]) for r in probe.records if r["kvol"] == 27}):
                ro, _ = roofline_from_timer(probe, meta_timer, want=k)
                if ro is not None:
                    others["%dx%d_k%d" % k[:3]] = {f: ro[f] for f in ("bound", "achieved", "peak", "unit", "frac", "launches", "avg_launch_us",
                                                                       "algorithmic_bytes_per_launch") if f in ro}
            res["roofline_probe_step_by_kernel"] = others
            rows_of = {}
            for r in meta_timer.records:
                if r["kvol"] == 27:
                    rows_of.setdefault("%dx%d_k27" % (r["cin"], r["cout"]), set()).add(r["n_out"])
            res["conv_rows_by_kernel"] = {k: sorted(v) for k, v in rows_of.items()}   # identifies the launches in a rocprofv3 trace
            if api is not None:
                res["roofline_by_kernel"] = roofline_by_kernel(probe, meta_timer, api)
                res["roofline_by_kernel_note"] = (
                    "every kernel class of ONE step that takes >= 30 us: us_per_step = HIP-event time (convolutions: events "
                    "inside libdf3d_hip.so around each launch, untimed probe step of the timed configuration; others: events "
                    "around the C-ABI entry on its stream, one untimed step on the per-module path), achieved = SURVEY 8(d) "
                    "algorithmic bytes or flops / that time; mfma peak = dense bf16 / 3 (split precision spends three products)")
        if world == 1 and not args.no_cpu_baseline and args.workload in ("cp_fusion", "cp_lidar"):
            res["cpu_baseline"] = cpu_baseline(wl, args.cpu_sweeps)
        if precision_evidence is not None:
            res["precision_evidence"] = precision_evidence
        if "fp32_detect" in extra:
            # the companions (three bf16 parts; exact-fp32 MFMA) inside a field the driver's record keeps
            res["dtype"] = "%s; same step with 3 bf16 parts / 6 products %.2f ms; with exact-fp32-MFMA convolutions %.2f ms" % (
                res["dtype"], per_step(extra.get("split3_detect", 0.0)), per_step(extra["fp32_detect"]))
        if (world == 1 and args.side_configs and not protocol and stage == "detect" and args.workload == "cp_fusion"
                and not args.no_extra_passes):
            note("side configs")
            if hasattr(wl, "close"):
                wl.close()
            wl = None
            import gc
            gc.collect()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()                                # the other workloads get the device memory
            res["configs"] = side_configs(args)                     # LAST key: the driver's record keeps the tail of the line
        print(json.dumps(res))
        sys.stdout.flush()
    if hasattr(wl, "close"):
        wl.close()
    if D.is_dist():
        barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
