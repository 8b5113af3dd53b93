import os, sys, torch
sys.path.insert(0, "/root/repo/3d-dual-fusion_amd")
from dualfusion import ops
dev = torch.device("cuda:0")
rows = 62268
g = torch.Generator(device=dev).manual_seed(0)
w1 = torch.randn(1024, 128, device=dev, generator=g) * 0.05
w2 = torch.randn(128, 1024, device=dev, generator=g) * 0.05
x = torch.randn(rows, 128, device=dev, generator=g)
pk = ops.ffn_pack(w1, w2)
b1 = torch.zeros(1024, device=dev); b2 = torch.zeros(128, device=dev)
lw = torch.ones(128, device=dev); lb = torch.zeros(128, device=dev)
for _ in range(3):
    ops.ffn_fused(x, pk, b1, b2, 1024, residual=x, ln_weight=lw, ln_bias=lb)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.ffn_fused(x, pk, b1, b2, 1024, residual=x, ln_weight=lw, ln_bias=lb)
e1.record(); torch.cuda.synchronize()
print("FFN_DBG=%s CFG=%s: %.1f us" % (os.environ.get("DF3D_FFN_DBG", "0"), os.environ.get("DF3D_FFN_CFG", "rule"), e0.elapsed_time(e1) * 50))
