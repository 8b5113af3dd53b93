"""Probe (GPU): what the library GEMMs of the training formulation's image-side chain cost in fp32 and as three bf16
products with fp32 output (hi*hi + hi*lo + lo*hi), on the shapes of the CenterPoint adapter: [6*150*267 = 240300, 256] x [256, 256]."""
import time
import torch

dev = torch.device("cuda:0")
M, K, N = 240300, 256, 256
x = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev) * 0.05
g = torch.randn(M, N, device=dev)


def timed(f, n=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def split(t):
    hi = t.bfloat16()
    lo = (t - hi.float()).bfloat16()
    return hi, lo


ref = (x.double() @ w.double().t())
y32 = x @ w.t()
print("fp32 fwd  x@w.t   %.3f ms  err %.2e" % (timed(lambda: x @ w.t()), ((y32 - ref).abs().max() / ref.abs().max()).item()))
print("fp32 dgrad g@w    %.3f ms" % timed(lambda: g @ w))
print("fp32 wgrad g.t@x  %.3f ms" % timed(lambda: g.t() @ x))
xh, xl = split(x)
wh, wl = split(w)
gh, gl = split(g)
try:
    def f3():
        y = torch.mm(xh, wh.t(), out_dtype=torch.float32)
        y += torch.mm(xh, wl.t(), out_dtype=torch.float32)
        y += torch.mm(xl, wh.t(), out_dtype=torch.float32)
        return y
    y3 = f3()
    print("bf16x3 (3 mm, out fp32) %.3f ms err %.2e" % (timed(f3), ((y3 - ref).abs().max() / ref.abs().max()).item()))
except Exception as e:  # noqa
    print("mm out_dtype unsupported:", repr(e)[:200])
try:
    xc = torch.cat([xh, xh, xl], 1).contiguous()
    wc = torch.cat([wh, wl, wh], 1).contiguous()
    fc = lambda: torch.mm(xc, wc.t(), out_dtype=torch.float32)
    yc = fc()
    print("bf16x3 (K-concat, out fp32) %.3f ms err %.2e" % (timed(fc), ((yc - ref).abs().max() / ref.abs().max()).item()))
    # wgrad: g^T x over M
    gc = torch.cat([gh, gh, gl], 0)            # [3M, N]
    xc2 = torch.cat([xh, xl, xh], 0)           # [3M, K]
    fw = lambda: torch.mm(gc.t(), xc2, out_dtype=torch.float32)
    refw = g.double().t() @ x.double()
    yw = fw()
    print("bf16x3 wgrad (M-concat) %.3f ms err %.2e" % (timed(fw), ((yw - refw).abs().max() / refw.abs().max()).item()))
    yw32 = g.t() @ x
    print("fp32 wgrad err %.2e" % ((yw32 - refw).abs().max() / refw.abs().max()).item())
except Exception as e:  # noqa
    print("concat path failed:", repr(e)[:200])
print("split pass (x -> hi, lo) %.3f ms" % timed(lambda: split(x)))
yb = timed(lambda: xh @ wh.t())
print("bf16 single mm (bf16 out) %.3f ms" % yb)
