"""fp32 accuracy of the library ops the training path still uses (MIOpen conv2d, SDPA, matmul) against float64."""
import torch, torch.nn.functional as F
dev = torch.device("cuda:0")
torch.manual_seed(0)
def rel(a, b): return float((a.double().cpu() - b).abs().max() / b.abs().max())
def check(name, fn, *shapes):
    xs = [torch.randn(*s) for s in shapes]
    a = [x.clone().to(dev).requires_grad_(True) for x in xs]
    b = [x.clone().double().requires_grad_(True) for x in xs]
    ya, yb = fn(*a), fn(*b)
    g = torch.randn(*yb.shape)
    ya.backward(g.to(dev)); yb.backward(g.double())
    print("%-40s fwd %.2e  " % (name, rel(ya.detach(), yb.detach())) + "  ".join("g%d %.2e" % (i, rel(p.grad, q.grad)) for i, (p, q) in enumerate(zip(a, b))))
for B, C, O, H in ((2, 128, 128, 20), (2, 512, 128, 20), (4, 512, 128, 180), (4, 128, 10, 180)):
    check("conv2d 3x3 %d->%d @%d" % (C, O, H), lambda x, w: F.conv2d(x, w, padding=1), (B, C, H, H), (O, C, 3, 3))
check("conv1d 1x1 128->128", lambda x, w: F.conv1d(x, w), (2, 128, 400), (128, 128, 1))
def sdpa(q, k, v): return F.scaled_dot_product_attention(q, k, v)
check("sdpa 24 x 400", sdpa, (2, 8, 24, 16), (2, 8, 400, 16), (2, 8, 400, 16))
check("sdpa 200 x 32400", sdpa, (4, 8, 200, 16), (4, 8, 32400, 16), (4, 8, 32400, 16))
check("linear 128->384", lambda x, w: F.linear(x, w), (2, 400, 128), (384, 128))
check("matmul [240k,128]x[128,128]", lambda x, w: x @ w, (240000, 128), (128, 128))
check("bmm", lambda x, w: torch.bmm(x, w), (24, 128, 256), (24, 256, 22400))
check("layer_norm", lambda x, w: F.layer_norm(x, (128,), w), (2, 24, 128), (128,))
check("batch_norm2d", lambda x, w: F.batch_norm(x, None, None, w, None, True), (4, 128, 180, 180), (128,))
check("group_norm", lambda x, w: F.group_norm(x, 32, w), (6, 128, 32, 56), (128,))
