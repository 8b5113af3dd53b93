#!/usr/bin/env python3
"""A/B of the weight-stationary conv kernel (spconv_ws.h) against the round-2 kernels on the real nuScenes stage geometry:
per backbone shape the time per launch with DF3D_CONV_HALO=0 / 1 (and both tile heights), and the max deviation between them
and from the exact-fp32 kernel.  usage: halo_probe.py [iters]"""
import os
import sys

os.environ["DF3D_EXECUTOR"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

from dualfusion import ops, synth  # noqa: E402
from dualfusion.pipeline import CenterPointHotPath  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = CenterPointHotPath().eval().to(dev)
pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
with torch.no_grad():
    feats, coors = model.voxelize(pts)
    x1, x2, x3, x4 = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)


def timeit(fn):
    for _ in range(3):
        y = fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        y = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters, y


cases = [("conv2 subm 32->32", x2, model.backbone.conv2[3].conv1), ("conv3 subm 64->64", x3, model.backbone.conv3[3].conv1),
         ("conv4 subm 128->128", x4, model.backbone.conv4[3].conv1),
         ("conv3 down 32->64", x2, model.backbone.conv3[0]), ("conv4 down 64->128", x3, model.backbone.conv4[0])]
for name, x, conv in cases:
    rb = x.find_indice_pair(conv.indice_key)
    if rb is None:                                   # strided layer: build its rulebook
        with torch.no_grad():
            y = conv(x)
        rb = y.find_indice_pair(conv.indice_key) or x.find_indice_pair(conv.indice_key)
    if rb is None:
        print(name, "no rulebook under key", conv.indice_key)
        continue
    cin, cout = conv.in_channels, conv.out_channels
    w = conv.weight.detach().view(-1, cin, cout).contiguous()
    f = x.features.contiguous()
    n_out = rb.nbr.shape[1]
    R = int((rb.nbr >= 0).sum())
    packed, fs = ops.conv_pack_weights(w), ops.split_rows(f)
    y32 = ops.sparse_conv_fused(f, w, rb.nbr, n_out, relu=True)
    abytes = R * cin * 4 + n_out * cout * 4 + 8 * R + w.numel() * 4
    res = {}
    for tag, env in (("round2", {"DF3D_CONV_WS": "0"}), ("ws", {"DF3D_CONV_WS": "1"}), ("round2 again", {"DF3D_CONV_WS": "0"}), ("ws again", {"DF3D_CONV_WS": "1"})):
        os.environ.update(env)
        us, (ys, ysp) = timeit(lambda: ops.sparse_conv_split(fs, packed, rb.nbr, n_out, cin, cout, relu=True))
        res[tag] = (us, ys, ysp)
    os.environ.pop("DF3D_CONV_WS", None)
    base = res["round2"][1]
    line = "%-22s n_in %6d n_out %6d pairs %7d (%.1f/row) alg %.0f MB:" % (name, f.shape[0], n_out, R, R / n_out, abytes / 1e6)
    for tag, (us, ys, ysp) in res.items():
        d = float((ys - base).abs().max() / base.abs().max())
        d32 = float((ys - y32).abs().max() / y32.abs().max())
        same_split = bool(torch.equal(ops.split_rows(ys), ysp)) if ysp is not None else None
        line += "  %s %.1f us (%.2f of HBM roof; vs round2 %.1e, vs fp32 %.1e, split rows ok %s)" % (
            tag, us, abytes / us / 1e6 / 8.0, d, d32, same_split)
    print(line)
