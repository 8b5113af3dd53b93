#!/usr/bin/env python3
"""wgrad_split3_kernel<*, *, NP> (NP = 1 bf16, 2 fp16 pairs, 3 bf16 triples) on the SubM layers of a nuScenes sweep (conv2 / conv3 /
conv4 of the CenterPoint backbone, batch 1 and batch 4 row counts) and on a dense 3 x 3 map: microseconds per call."""
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


def timeit(fn, iters=20):
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


B = int(os.environ.get("B", "1"))
model = CenterPointHotPath().eval().to(dev)
pts = [torch.from_numpy(synth.nusc_sweep(seed=s)).to(dev) for s in range(B)]
with torch.no_grad():
    feats, coors = model.voxelize(pts)
    x1, x2, x3, x4 = model.backbone._stem(feats, coors, B, model.grid_size_xyz)
for stage, x in (("conv4", x4), ("conv3", x3), ("conv2", x2)):
    blk = getattr(model.backbone, stage)[3]
    rb = x.find_indice_pair(blk.conv1.indice_key)
    f = x.features.contiguous()
    g = torch.randn_like(f) * 1e-3
    os.environ["DF3D_WGRAD"] = "1"
    t1, ref = timeit(lambda: ops.sparse_conv_grad_filters(f, g, rb.nbr))
    os.environ["DF3D_WGRAD"] = "3"
    t3, y3 = timeit(lambda: ops.sparse_conv_grad_filters(f, g, rb.nbr))
    os.environ.pop("DF3D_WGRAD")
    sc = ops.rows_pow2_scale(g)
    t2, y2 = timeit(lambda: ops.sparse_conv_grad_filters(f, g, rb.nbr, grad_scale=sc))
    tb, yb = timeit(lambda: ops.sparse_conv_grad_filters(f, g, rb.nbr, bf16=True))
    R = int((rb.nbr >= 0).sum())
    err = lambda y: float((y.double() - ref.double()).abs().max() / ref.abs().max())
    print("%s %d ch, %d rows, %d pairs: fp32 %.0f us | NP=3 %.0f us (%.1e) | NP=2 %.0f us (%.1e) | NP=1 %.0f us (%.1e); gather floor %.0f us at 4 TB/s"
          % (stage, f.shape[1], f.shape[0], R, t1, t3, err(y3), t2, err(y2), tb, err(yb), 2.0 * R * f.shape[1] * 4 / 4e6), flush=True)
