"""MSDA backward at the TransFusion training shape (24 images x ~10 k padded queries, 8 heads x 4 points, 128 ch on 112 x 200):
time per call, with the reference points of a third of the queries at (0, 0) like the unseen voxels' and the padded rows'."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch
from dualfusion import ops
dev = torch.device("cuda:0")
N, Lq, M, D, P, H, W = int(os.environ.get("N", 24)), int(os.environ.get("LQ", 10000)), 8, 16, 4, 112, 200
g = torch.Generator(device="cpu").manual_seed(0)
value = torch.randn(N, H * W, M, D, generator=g).to(dev)
ref = torch.rand(N, Lq, 1, 1, 1, 2, generator=g)
HOT = float(os.environ.get('HOT', '0.3333'))
if HOT > 0: ref[:, int(Lq * (1 - HOT)):] = 0.0
loc = (ref + torch.randn(N, Lq, M, 1, P, 2, generator=g) * 0.01).to(dev).contiguous()
aw = torch.softmax(torch.randn(N, Lq, M, P, generator=g), -1).view(N, Lq, M, 1, P).to(dev).contiguous()
go = torch.randn(N, Lq, M * D, generator=g).to(dev)
go[:, Lq * 5 // 6:] = 0.0                          # padded rows: no upstream gradient
shp = torch.tensor([[H, W]], dtype=torch.long, device=dev)
ls = torch.zeros(1, dtype=torch.long, device=dev)
def run():
    return ops.ms_deform_attn_backward(value, shp, ls, loc, aw, go)
for _ in range(3): out = run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): out = run()
b.record(); torch.cuda.synchronize()
print("hot=%s scope=%s  %.1f us per call; grad_value checksum %.6e" % (os.environ.get("HOT", "0.33"), os.environ.get("DF3D_MSDA_ATOMIC_SCOPE", "agent"), a.elapsed_time(b) * 100, float(out[0].double().sum())))
