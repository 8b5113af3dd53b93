"""BatchNorm (train mode) backward when the upstream gradient is nearly constant over the rows (what a heat-map loss sends):
MIOpen's, the row kernels' (csrc/bnrows.hip) and CPU torch's fp32 results against float64."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch, torch.nn.functional as F
from dualfusion import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def rel(a, b): return float((a.double().cpu() - b).abs().max() / b.abs().max())
for n, C, noise, xmean in ((800, 128, 1.0, 0.0), (800, 128, 1e-3, 0.0), (800, 128, 1e-3, 3.0), (40000, 64, 1e-3, 3.0), (129600, 128, 1e-2, 1.0)):
    x = torch.randn(n, C) + xmean
    dy = 1.0 + noise * torch.randn(n, C)
    w = 1 + 0.1 * torch.randn(C)
    res = {}
    for tag, dt, dv in (("f64", torch.float64, "cpu"), ("cpu32", torch.float32, "cpu"), ("miopen1d", torch.float32, dev), ("rows", torch.float32, dev),
                        ("miopen2d", torch.float32, dev)):
        xx = x.clone().to(dv, dt).requires_grad_(True); ww = w.clone().to(dv, dt).requires_grad_(True); bb = torch.zeros(C, device=dv, dtype=dt, requires_grad=True)
        if tag == "rows":
            bn = torch.nn.BatchNorm1d(C).to(dev).train()
            with torch.no_grad(): bn.weight.copy_(w)
            y = ops.batch_norm_rows(bn, xx, relu=False)
            y.backward(dy.to(dev))
            res[tag] = (y.detach(), xx.grad, bn.weight.grad)
            continue
        if tag == "miopen2d":
            x4 = xx.t().reshape(1, C, n // 20, 20)
            y = F.batch_norm(x4, None, None, ww, bb, True, 0.1, 1e-5)
            y.backward(dy.t().reshape(1, C, n // 20, 20).to(dv, dt))
            res[tag] = (y.detach().reshape(C, n).t(), xx.grad, ww.grad)
            continue
        y = F.batch_norm(xx, None, None, ww, bb, True, 0.1, 1e-5)
        y.backward(dy.to(dv, dt))
        res[tag] = (y.detach(), xx.grad, ww.grad)
    ref = [t.double().cpu() for t in res["f64"]]
    print("n %d C %d noise %g xmean %g |dx| %.2e" % (n, C, noise, xmean, float(ref[1].abs().max())))
    for tag in ("cpu32", "miopen1d", "miopen2d", "rows"):
        print("   %-9s y %.2e  dx %.2e  dw %.2e" % (tag, rel(res[tag][0], ref[0]), rel(res[tag][1], ref[1]), rel(res[tag][2], ref[2])))
