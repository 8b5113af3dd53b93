"""Final convolutions of the head's 36 branches on one 180 x 180 map (csrc/headconv.hip): launch time per workgroup order.
DF3D_HEADFINAL_ORDER = 0 (tile-fastest 2-D grid, rounds 1-2), 1 (branch-fastest, branches padded to 8: the default), 2
(branch-fastest, unpadded).  Run once per value (the switch is read once per process)."""
import os
import sys
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "3d-dual-fusion_amd"))
from dualfusion import ops  # noqa: E402

dev = torch.device("cuda:0")
B, H, W, G = 1, 180, 180, 36
g = torch.Generator(device=dev).manual_seed(0)
acts = torch.randn(B * H * W, G * 64, device=dev, generator=g)
split = ops.split_rows(acts)
w = torch.randn(G, 9, 64, 4, device=dev, generator=g) * 0.05
b = torch.randn(G, 4, device=dev, generator=g)
cols = torch.tensor([[2 * i, 2] for i in range(G)], dtype=torch.int32, device=dev)
pk = ops.head_final_pack(w) if os.environ.get("HF_PACKED", "1") == "1" and os.environ.get("DF3D_HEADFINAL", "m")[:1] != "v" else None
out = ops.head_final_conv(split, B, H, W, w, b, cols, 72, packed=pk)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    ops.head_final_conv(split, B, H, W, w, b, cols, 72, packed=pk)
e0.record()
N = 50
for _ in range(N):
    ops.head_final_conv(split, B, H, W, w, b, cols, 72, packed=pk)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / N
print("kernel %s packed %s order %s: %.1f us  (%.0f GB/s of the %d MB of activations)  checksum %.6f" % (
    os.environ.get("DF3D_HEADFINAL", "mfma"), pk is not None, os.environ.get("DF3D_HEADFINAL_ORDER", "2"), us, split.numel() / us / 1e3, split.numel() >> 20, float(out.double().sum())))
