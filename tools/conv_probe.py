#!/usr/bin/env python3
"""Micro-benchmark of one sparse-conv layer at nuScenes size (for rocprofv3 / PMC runs).
usage: conv_probe.py [stage=conv4|conv3] [iters]"""
import os
import sys
import time

os.environ["DF3D_EXECUTOR"] = "0"      # the probe needs the per-module rulebooks (indice_dict)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

from dualfusion import ops, synth  # noqa: E402
from dualfusion.pipeline import CenterPointHotPath  # noqa: E402

stage = sys.argv[1] if len(sys.argv) > 1 else "conv4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = CenterPointHotPath().eval().to(dev)
pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
with torch.no_grad():
    feats, coors = model.voxelize(pts)
    x1, x2, x3, x4 = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)
x = {"conv4": x4, "conv3": x3, "conv2": x2}[stage]
blk = getattr(model.backbone, stage)[3]
rb = x.find_indice_pair(blk.conv1.indice_key)
C = x.features.shape[1]
w = blk.conv1.weight.detach().view(-1, C, C).contiguous()
f = x.features.contiguous()
R = int((rb.nbr >= 0).sum())
n = f.shape[0]
tiles = rb.tiles(C, C) if os.environ.get('DF3D_BALANCE', '1') == '1' else None
def timeit(fn):
    for _ in range(3):
        y = fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        y = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters, y


fl = 2.0 * R * C * C
abytes = R * C * 4 + n * C * 4 + rb.nbr.numel() * 4 + w.numel() * 4
us, y32 = timeit(lambda: ops.sparse_conv_fused(f, w, rb.nbr, n, relu=True, tiles=tiles))
print("stage %s N=%d C=%d R=%d pairs/row=%.1f : fp32 kernel %.1f us/launch, %.1f TF useful (%.1f%% of 157.3 fp32 MFMA)"
      % (stage, n, C, R, R / n, us, fl / us / 1e6, fl / us / 1e6 / 157.3 * 100))
if ops.conv_split_supported(rb.nbr.shape[0], C, C):
    packed = ops.conv_pack_weights(w)
    fs = ops.split_rows(f)
    us, (ys, _) = timeit(lambda: ops.sparse_conv_split(fs, packed, rb.nbr, n, C, C, relu=True, tiles=tiles))
    err = float((ys - y32).abs().max() / y32.abs().max())
    print("  split-precision kernel (%s) %.1f us/launch, %.1f TF useful, %.2f TB/s algorithmic (%.1f%% of 8 TB/s HBM), "
          "max |diff| vs fp32 kernel %.2e of scale" % (os.environ.get("DF3D_SPLIT_KERNEL", "default"), us,
                                                      fl / us / 1e6, abytes / us / 1e6, abytes / us / 1e6 / 8 * 100,
                                                      err))
    us, _ = timeit(lambda: ops.split_rows(f))
    print("  split_rows pass %.1f us" % us)
# ---- tile balance statistics (rows cut into 256 equal tiles, as the pair kernel does)
import math
cnt_row = (rb.nbr >= 0).sum(0).float()
for ntile in (256, 512):
    TM = (math.ceil(n / ntile) + 3) // 4 * 4
    nt = math.ceil(n / TM)
    pad = torch.zeros(nt * TM, device=dev)
    pad[:n] = cnt_row
    per_tile = pad.view(nt, TM).sum(1)
    # chunks per tile: sum_k ceil(n_k/16)
    valid = torch.zeros((rb.nbr.shape[0], nt * TM), device=dev)
    valid[:, :n] = (rb.nbr >= 0).float()
    nk = valid.view(rb.nbr.shape[0], nt, TM).sum(2)
    chunks = torch.ceil(nk / 16).sum(0)
    print("tiles=%d TM=%d pairs/tile mean %.0f max %.0f (max/mean %.2f); chunks/tile mean %.1f max %.0f (max/mean %.2f); slot efficiency %.2f"
          % (nt, TM, per_tile.mean(), per_tile.max(), per_tile.max() / per_tile.mean(), chunks.mean(), chunks.max(),
             chunks.max() / chunks.mean(), float(per_tile.sum() / (chunks.sum() * 16))))
