"""The dual-query layer's two feed-forward blocks (LayerNorm(x + W2 relu(W1 x)), d_model 128, d_ffn 1024) as one launch of
csrc/ffn.hip at the nuScenes query counts (2 x 31134 rows).  DF3D_FFN_CFG = NW*10 + RT picks the workgroup shape (read once
per process): run once per value."""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "3d-dual-fusion_amd"))
from dualfusion import ops  # noqa: E402

dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 31134
g = torch.Generator(device=dev).manual_seed(0)
jobs = []
for i in range(2):
    w1 = torch.randn(1024, 128, device=dev, generator=g) * 0.05
    w2 = torch.randn(128, 1024, device=dev, generator=g) * 0.05
    x = torch.randn(rows, 128, device=dev, generator=g)
    jobs.append(dict(x=x, packed=ops.ffn_pack(w1, w2), b1=torch.randn(1024, device=dev, generator=g) * 0.1,
                     b2=torch.randn(128, device=dev, generator=g) * 0.1, residual=x, ln_weight=torch.ones(128, device=dev),
                     ln_bias=torch.zeros(128, device=dev), eps=1e-5, w1=w1, w2=w2))
outs = ops.ffn_fused_jobs(jobs, 1024)
j = jobs[0]
h = torch.relu(j["x"].double() @ j["w1"].double().t() + j["b1"].double())
ref = torch.nn.functional.layer_norm(j["x"].double() + h @ j["w2"].double().t() + j["b2"].double(), (128,))
err = float((outs[0].double() - ref).abs().max())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    ops.ffn_fused_jobs(jobs, 1024)
e0.record()
N = 30
for _ in range(N):
    ops.ffn_fused_jobs(jobs, 1024)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / N
print("cfg %s: %.1f us for 2 x %d rows (%.0f TF of fp32-grade products, x3 on the matrix cores)  max |err| vs float64 %.2e" % (
    os.environ.get("DF3D_FFN_CFG", "81"), us, rows, 2 * rows * 128 * 1024 * 4 / us / 1e6, err))
