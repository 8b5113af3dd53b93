#!/usr/bin/env python3
"""Filter-gradient kernels on the conv4 / conv3 layers of a nuScenes sweep and on the feed-forward weight gradients of the
adapter: wgrad_f32_kernel (DF3D_WGRAD=1) against wgrad_split3_kernel (3), and torch's x^T g GEMM against df3d_rows_grad_weights."""
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
        y = fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        y = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters, y


model = CenterPointHotPath().eval().to(dev)
pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
with torch.no_grad():
    feats, coors = model.voxelize(pts)
    x1, x2, x3, x4 = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)
for stage, x in (("conv4", x4), ("conv3", x3), ("conv2", x2)):
    blk = getattr(model.backbone, stage)[3]
    rb = x.find_indice_pair(blk.conv1.indice_key)
    f = x.features.contiguous()
    g = torch.randn_like(f) * 1e-3
    res = {}
    for mode in ("1", "3"):
        os.environ["DF3D_WGRAD"] = mode
        res[mode] = timeit(lambda: ops.sparse_conv_grad_filters(f, g, rb.nbr))
    os.environ.pop("DF3D_WGRAD")
    sc = ops.rows_pow2_scale(g)
    t2, y2 = timeit(lambda: ops.sparse_conv_grad_filters(f, g, rb.nbr, grad_scale=sc))
    print("   two-part kernel (fp16 pairs, block-scaled gradient) %.0f us, max diff %.1e of scale" % (
        t2, float((y2.double() - res["1"][1].double()).abs().max() / res["1"][1].abs().max())), flush=True)
    R = int((rb.nbr >= 0).sum())
    ref = torch.zeros_like(res["1"][1], dtype=torch.float64)
    d = float((res["3"][1].double() - res["1"][1].double()).abs().max() / res["1"][1].abs().max())
    print("%s K=27 %d ch, %d rows, %d pairs: fp32 kernel %.0f us, three-part kernel %.0f us (%.0f TFLOP/s of pairs), max diff %.1e of scale"
          % (stage, f.shape[1], f.shape[0], R, res["1"][0], res["3"][0], 2.0 * R * f.shape[1] ** 2 / res["3"][0] / 1e6, d), flush=True)
for n, cin, cout in ((32034, 128, 1024), (32034, 1024, 128), (240300, 256, 128), (240300, 128, 128), (240300, 4, 256)):
    x = torch.randn(n, cin, device=dev)
    g = torch.randn(n, cout, device=dev) * 1e-3
    t_t, ref = timeit(lambda: x.t() @ g)
    t_k, got = timeit(lambda: ops.rows_grad_weights(x, g))
    sc = ops.rows_pow2_scale(g)
    t_2, got2 = timeit(lambda: ops.rows_grad_weights(x, g, g_scale=sc))
    print("x^T g, %d rows, %d x %d: torch %.0f us, df3d_rows_grad_weights %.0f us, two-part %.0f us, max diff %.1e / %.1e of scale"
          % (n, cin, cout, t_t, t_k, t_2, float((got - ref).abs().max() / ref.abs().max()),
             float((got2 - ref).abs().max() / ref.abs().max())), flush=True)
for cin, cout, H in ((128, 128, 180), (256, 256, 90), (256, 128, 180), (512, 64, 180), (64, 2304, 180), (128, 256, 90)):
    nbr, _, _ = ops.conv2d_neighbors(1, H, H, 3, 3, 1, 1, False, dev)
    n = nbr.shape[1]
    f = torch.randn(n, cin, device=dev)
    g = torch.randn(n, cout, device=dev) * 1e-3
    res = {}
    for mode in ("1", "3"):
        os.environ["DF3D_WGRAD"] = mode
        res[mode] = timeit(lambda: ops.sparse_conv_grad_filters(f, g, nbr), iters=5)
    os.environ.pop("DF3D_WGRAD")
    sc = ops.rows_pow2_scale(g)
    t2, y2 = timeit(lambda: ops.sparse_conv_grad_filters(f, g, nbr, grad_scale=sc), iters=5)
    print("dense 3x3 %d -> %d on %d x %d: fp32 kernel %.0f us, three-part kernel %.0f us, two-part %.0f us, max diff %.1e / %.1e of scale"
          % (cin, cout, H, H, res["1"][0], res["3"][0], t2, float((res["3"][1] - res["1"][1]).abs().max() / res["1"][1].abs().max()),
             float((y2 - res["1"][1]).abs().max() / res["1"][1].abs().max())), flush=True)
