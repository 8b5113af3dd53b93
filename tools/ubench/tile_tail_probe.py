"""Does the K = 27 64 -> 64 layer pay for a second round of workgroups?  128-row tiles, two 8-wave workgroups per CU = 512 slots;
a nuScenes sweep has 508 .. 520 tiles.  Times the layer on its first n rows for n around 512 tiles (tools/conv_probe.py set-up)."""
import os
import sys

os.environ["DF3D_EXECUTOR"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

from dualfusion import ops, synth  # noqa: E402
from dualfusion.pipeline import CenterPointHotPath  # noqa: E402

stage = sys.argv[1] if len(sys.argv) > 1 else "conv3"
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = CenterPointHotPath().eval().to(dev)
pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
with torch.no_grad():
    feats, coors = model.voxelize(pts)
    x1, x2, x3, x4 = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)
x = {"conv4": x4, "conv3": x3, "conv2": x2}[stage]
blk = getattr(model.backbone, stage)[3]
rb = x.find_indice_pair(blk.conv1.indice_key)
C = x.features.shape[1]
w = blk.conv1.weight.detach().view(-1, C, C).contiguous()
packed = ops.conv_pack_weights(w)
fs = ops.split_rows(x.features.contiguous())
N = rb.nbr.shape[1]
print("%s: %d rows = %d tiles of 128" % (stage, N, (N + 127) // 128))
for tiles in (256, 384, 448, 500, 508, 512, 513, 516, 520, 560, 640, 768, (N + 127) // 128):
    n = min(N, tiles * 128)
    nbr = rb.nbr[:, :n].contiguous()
    for _ in range(3):
        ops.sparse_conv_split(fs, packed, nbr, n, C, C, relu=True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(30):
        ops.sparse_conv_split(fs, packed, nbr, n, C, C, relu=True)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) * 1e3 / 30
    print("  %4d tiles (%6d rows): %6.1f us  = %.3f us per tile" % ((n + 127) // 128, n, us, us / ((n + 127) // 128)))
