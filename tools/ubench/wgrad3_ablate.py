#!/usr/bin/env python3
"""wgrad_split3_kernel on conv4 / conv3 of a nuScenes sweep with its phases switched off one at a time (DF3D_W3_DBG: 1 no
atomics, 2 no MFMAs, 4 no split + LDS stores) and with other workgroup counts (DF3D_W3_WGS)."""
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


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters


model = CenterPointHotPath().eval().to(dev)
pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
with torch.no_grad():
    feats, coors = model.voxelize(pts)
    x1, x2, x3, x4 = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)
os.environ["DF3D_WGRAD"] = "3"
for stage, x in (("conv4", x4), ("conv3", x3)):
    blk = getattr(model.backbone, stage)[3]
    rb = x.find_indice_pair(blk.conv1.indice_key)
    f = x.features.contiguous()
    g = torch.randn_like(f) * 1e-3
    out = []
    for dbg in (0, 1, 2, 4, 6, 7):
        os.environ["DF3D_W3_DBG"] = str(dbg)
        out.append("dbg %d: %.0f" % (dbg, timeit(lambda: ops.sparse_conv_grad_filters(f, g, rb.nbr))))
    os.environ["DF3D_W3_DBG"] = "0"
    print(stage, f.shape, "  ".join(out), "us", flush=True)
    out = []
    for wgs in (256, 512, 1024, 2048, 4096):
        os.environ["DF3D_W3_WGS"] = str(wgs)
        out.append("wgs %d: %.0f" % (wgs, timeit(lambda: ops.sparse_conv_grad_filters(f, g, rb.nbr))))
    os.environ.pop("DF3D_W3_WGS")
    print(stage, "  ".join(out), "us", flush=True)
