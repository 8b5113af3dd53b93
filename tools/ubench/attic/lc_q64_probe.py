#!/usr/bin/env python3
"""A/B of the loader / consumer kernel's matrix-wave tiling (DF3D_LC_Q64=0: 32 rows x 128 columns per wave, 1: 64 x 64) on
every layer shape the CenterPoint step runs on it: conv4's K = 27 SubM layer (nuScenes sweep), the 3 x 3 neck layers on the
180 x 180 map, the head's 64 -> 36 x 64 middle convolutions.  Same values bit for bit (same summation order), time per launch."""
import os
import sys

os.environ["DF3D_EXECUTOR"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

from dualfusion import ops, synth  # noqa: E402
from dualfusion.pipeline import CenterPointHotPath  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30


def timeit(fn):
    for _ in range(3):
        y = fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        y = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters, y


def ab(name, fn):
    out = []
    ys = []
    for q in ("0", "1", "0", "1"):
        os.environ["DF3D_LC_Q64"] = q
        us, y = timeit(fn)
        out.append("%s: %.1f" % (q, us))
        ys.append(y)
    same = all(torch.equal(a, b) for a, b in zip(ys[0], ys[1]) if a is not None)
    print("%-34s %s us   bit-identical %s" % (name, "  ".join(out), same), flush=True)


model = CenterPointHotPath().eval().to(dev)
pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
with torch.no_grad():
    feats, coors = model.voxelize(pts)
    x1, x2, x3, x4 = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)
for stage, x in (("conv4", x4), ("conv3", x3)):
    blk = getattr(model.backbone, stage)[3]
    rb = x.find_indice_pair(blk.conv1.indice_key)
    C = x.features.shape[1]
    if C != 128:
        continue
    w = blk.conv1.weight.detach().view(-1, C, C).contiguous()
    pk, fs, n = ops.conv_pack_weights(w), ops.split_rows(x.features.contiguous()), x.features.shape[0]
    ab("%s K=27 %d->%d, %d rows" % (stage, C, C, n),
       lambda: ops.sparse_conv_split(fs, pk, rb.nbr, n, C, C, relu=True))
g = torch.Generator(device=dev).manual_seed(0)
nbr, _, _ = ops.conv2d_neighbors(1, 180, 180, 3, 3, 1, 1, False, dev)
n = nbr.shape[1]
for cin, cout in ((128, 128), (256, 128)):
    x = ops.split_rows(torch.randn(n, cin, device=dev, generator=g))
    pk = ops.conv_pack_weights(torch.randn(9, cin, cout, device=dev, generator=g) * 0.05)
    sc = torch.ones(cout, device=dev)
    ab("neck 3x3 %d->%d 180x180" % (cin, cout),
       lambda: ops.sparse_conv_split(x, pk, nbr, n, cin, cout, scale=sc, shift=sc, relu=True))
G, cin, cout = 18, 64, 128
x = ops.split_rows(torch.randn(n, cin, device=dev, generator=g))
pk = torch.cat([ops.conv_pack_weights(torch.randn(9, cin, cout, device=dev, generator=g) * 0.05) for _ in range(G)])
sc = torch.ones(G * cout, device=dev)
ab("head 3x3 64->18x128 180x180",
   lambda: ops.conv_rows_split(x, cin, 0, pk, cout, G, nbr, n, None, sc, sc, relu=True, want_out=False, want_split=True))
