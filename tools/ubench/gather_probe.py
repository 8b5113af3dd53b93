"""Gather cost of the backbone's K = 27 layers by itself, three lane mappings (tools/ubench/gather_probe.hip).
Build first: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench/gather_probe.hip -o tools/ubench/gather_probe.so"""
import ctypes
import os
import sys

os.environ["DF3D_EXECUTOR"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

from dualfusion import ops, synth  # noqa: E402
from dualfusion.pipeline import CenterPointHotPath  # noqa: E402

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gather_probe.so"))
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = CenterPointHotPath().eval().to(dev)
pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
with torch.no_grad():
    feats, coors = model.voxelize(pts)
    x1, x2, x3, x4 = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)
zero = torch.zeros(4096, dtype=torch.uint8, device=dev)
for name, x, conv in (("conv2 32ch", x2, model.backbone.conv2[3].conv1), ("conv3 64ch", x3, model.backbone.conv3[3].conv1),
                      ("conv4 128ch", x4, model.backbone.conv4[3].conv1)):
    rb = x.find_indice_pair(conv.indice_key)
    cin = conv.in_channels
    fs = ops.split_rows(x.features.contiguous())
    n_out = rb.nbr.shape[1]
    R = int((rb.nbr >= 0).sum())
    line = "%-12s rows %6d pairs %7d gathered %.0f MB:" % (name, n_out, R, R * cin * 4 / 1e6)
    for grid in (256, 512, 1024):
        for mode in (0, 1, 2):
            sink = torch.zeros(grid * 8, dtype=torch.int32, device=dev)
            call = lambda: lib.gather_probe(ctypes.c_void_p(fs.data_ptr()), ctypes.c_void_p(rb.nbr.data_ptr()), n_out, 27, cin, mode, grid,  # noqa: E731
                                            ctypes.c_void_p(zero.data_ptr()), ctypes.c_void_p(sink.data_ptr()),
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            for _ in range(3):
                assert call() == 0
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                call()
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) * 1e3 / 20
            line += "  g%d/m%d %.1f us" % (grid, mode, us)
    print(line)
