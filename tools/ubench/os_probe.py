#!/usr/bin/env python3
"""Timing of the output-stationary split-precision conv kernel on the shapes that dominate the neck / head / backbone
(HIP events around repeated launches; tuning aid for DF3D_OS_CFG / DF3D_OS_WIDE / kernel variants).
usage: os_probe.py [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

from dualfusion import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


def dense(cin, cout, H, W, k=3, stride=1, groups=1, tag=""):
    nbr, Ho, Wo = ops.conv2d_neighbors(1, H, W, k, k, stride, k // 2, False, dev)
    n_in, n_out = H * W, Ho * Wo
    f = torch.randn(n_in, cin, device=dev)
    fs = ops.split_rows(f)
    w = torch.randn(k * k, cin, cout, device=dev) * 0.05
    if groups == 1:
        packed = ops.conv_pack_weights(w)
        us = timeit(lambda: ops.sparse_conv_split(fs, packed, nbr, n_out, cin, cout, relu=True))
    else:
        packed = torch.cat([ops.conv_pack_weights(w) for _ in range(groups)])
        us = timeit(lambda: ops.conv_rows_split(fs, cin, 0, packed, cout, groups, nbr, n_out, relu=True, want_out=False,
                                                want_split=True))
    fl = 2.0 * n_out * k * k * cin * cout * groups
    extra = ""
    if groups == 1 and ops.conv_bf16_supported(k * k, cin, cout):
        fb, pb = ops.rows_to_bf16(f), ops.conv_pack_weights_bf16(w)
        usb = timeit(lambda: ops.sparse_conv_bf16(fb, pb, nbr, n_out, cin, cout, relu=True))
        extra = " | bf16 %6.1f us (%5.1f TF)" % (usb, fl / usb / 1e6)
    print("%-28s %4d->%4dx%-2d rows %6d K=%2d : %7.1f us  %6.1f TF fp32-eq (%4.1f%% of the bf16/3 roof)%s" % (
        tag, cin, cout, groups, n_out, k * k, us, fl / us / 1e6, fl / us / 1e6 / 833 * 100, extra))


if os.environ.get("DF3D_PROBE_K1", "0") == "1":
    dense(128, 128, 180, 180, k=1, tag="1x1 (fixed cost probe)")
    dense(128, 128, 90, 90, k=1, tag="1x1 quarter rows")
    dense(128, 128, 90, 90, k=3, tag="3x3 quarter rows")
dense(128, 128, 180, 180, tag="neck block 1")
dense(256, 256, 90, 90, tag="neck block 2")
dense(256, 128, 180, 180, tag="neck first")
dense(512, 64, 180, 180, tag="head shared")
dense(64, 64, 180, 180, groups=36, tag="head mid")
if os.environ.get("DF3D_PROBE_SPARSE", "1") == "1":
    os.environ["DF3D_EXECUTOR"] = "0"
    from dualfusion import synth
    from dualfusion.pipeline import CenterPointHotPath
    model = CenterPointHotPath().eval().to(dev)
    pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
    with torch.no_grad():
        feats, coors = model.voxelize(pts)
        xs = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)
    for stage, x in (("conv4", xs[3]), ("conv3", xs[2]), ("conv2", xs[1])):
        blk = getattr(model.backbone, stage)[3]
        rb = x.find_indice_pair(blk.conv1.indice_key)
        C = x.features.shape[1]
        w = blk.conv1.weight.detach().view(-1, C, C).contiguous()
        fs = ops.split_rows(x.features.contiguous())
        n = x.features.shape[0]
        R = int((rb.nbr >= 0).sum())
        packed = ops.conv_pack_weights(w)
        us = timeit(lambda: ops.sparse_conv_split(fs, packed, rb.nbr, n, C, C, relu=True))
        ab = R * C * 4 + 2 * n * C * 4 + rb.nbr.numel() * 4 + w.numel() * 4
        fb, pb = ops.rows_to_bf16(x.features.contiguous()), ops.conv_pack_weights_bf16(w)
        usb = timeit(lambda: ops.sparse_conv_bf16(fb, pb, rb.nbr, n, C, C, relu=True))
        print("%-28s %4d->%4d    rows %6d K=27 pairs/row %.1f : %7.1f us  %6.1f TF useful, %.2f TB/s algorithmic | bf16 %6.1f us" % (
            "backbone " + stage, C, C, n, R / n, us, 2.0 * R * C * C / us / 1e6, ab / us / 1e6, usb))
