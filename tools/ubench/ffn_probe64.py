"""The 64-wide fused feed-forward block (ACTRv2 of the Voxel-RCNN tree: d_model 64, d_ffn 1024, two jobs per launch) at the bs = 8
query counts.  DF3D_FFN_CFG64 = 0 (8 waves x 16 rows) | 82 (8 x 32) | 44 (4 waves x 64 rows, one wave per SIMD), read once per
process: run once per value."""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "3d-dual-fusion_amd"))
from dualfusion import ops  # noqa: E402

dev = torch.device("cuda:0")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 160000
g = torch.Generator(device=dev).manual_seed(0)
jobs = []
for i in range(2):
    w1 = torch.randn(1024, 64, device=dev, generator=g) * 0.05
    w2 = torch.randn(64, 1024, device=dev, generator=g) * 0.05
    x = torch.randn(rows, 64, device=dev, generator=g)
    jobs.append(dict(x=x, packed=ops.ffn_pack(w1, w2), b1=torch.randn(1024, device=dev, generator=g) * 0.1,
                     b2=torch.randn(64, device=dev, generator=g) * 0.1, residual=x, ln_weight=torch.ones(64, device=dev),
                     ln_bias=torch.zeros(64, device=dev), eps=1e-5, w1=w1, w2=w2))
outs = ops.ffn_fused_jobs(jobs, 1024)
j = jobs[0]
n = min(rows, 20000)
h = torch.relu(j["x"][:n].double() @ j["w1"].double().t() + j["b1"].double())
ref = torch.nn.functional.layer_norm(j["x"][:n].double() + h @ j["w2"].double().t() + j["b2"].double(), (64,))
err = float((outs[0][:n].double() - ref).abs().max() / ref.abs().max())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    ops.ffn_fused_jobs(jobs, 1024)
e0.record()
N = 20
for _ in range(N):
    ops.ffn_fused_jobs(jobs, 1024)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / N
print("cfg64 %s: %.1f us for 2 x %d rows (%.0f TF of fp32-grade products)  max err vs float64 %.2e of scale  checksum %.6f" % (
    os.environ.get("DF3D_FFN_CFG64", "0"), us, rows, 2 * rows * 64 * 1024 * 4 / us / 1e6, err, float(outs[1].double().sum())))
