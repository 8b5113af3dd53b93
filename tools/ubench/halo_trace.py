#!/usr/bin/env python3
"""Phase times of the staged-range conv kernel per workgroup (library built with -DDF3D_OS_TRACE): prologue (neighbour
table + ranges), filter / range pipeline fill, the step loop, epilogue.  s_memtime ticks = 10 ns."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import numpy as np
import torch
from dualfusion import _lib, ops, synth
from dualfusion.pipeline import CenterPointHotPath

dev = torch.device("cuda:0")
torch.manual_seed(0)
os.environ["DF3D_EXECUTOR"] = "0"
model = CenterPointHotPath().eval().to(dev)
pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
with torch.no_grad():
    feats, coors = model.voxelize(pts)
    xs = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.df3d_debug_set_os_trace.argtypes = [ctypes.c_void_p]
for stage, x, conv in (("conv2", xs[1], model.backbone.conv2[3].conv1), ("conv3", xs[2], model.backbone.conv3[3].conv1),
                       ("conv4", xs[3], model.backbone.conv4[3].conv1)):
    rb = x.find_indice_pair(conv.indice_key)
    C, n = x.features.shape[1], x.features.shape[0]
    fs = ops.split_rows(torch.randn(n, C, device=dev))
    packed = ops.conv_pack_weights(torch.randn(27, C, C, device=dev) * 0.05)
    for rt in ("2", "4"):
        os.environ["DF3D_HALO_RT"] = rt
        nwg = (n + 64 * int(rt) - 1) // (64 * int(rt))
        tr = torch.zeros((nwg, 16, 8), dtype=torch.int64, device=dev)
        for rep in range(6):
            if rep == 5:
                lib.df3d_debug_set_os_trace(ctypes.c_void_p(tr.data_ptr()))
            ops.sparse_conv_split(fs, packed, rb.nbr, n, C, C, relu=True)
        torch.cuda.synchronize()
        lib.df3d_debug_set_os_trace(None)
        t = tr.cpu().numpy().astype(np.float64)[:, :4, :]
        t0 = t[:, :, 0].min()
        steps = t[:, 0, 5]
        ph = [(t[:, :, i + 1] - t[:, :, i]).mean() / 100.0 for i in range(4)]
        print("%s rt%s: %d workgroups, %.1f steps/tile | us: table %.2f  fill %.2f  loop %.2f (%.3f per step)  epilogue %.2f | "
              "workgroup start spread %.1f us, last end %.1f us" % (stage, rt, nwg, steps.mean(), ph[0], ph[1], ph[2],
                                                                    ph[2] / max(steps.mean(), 1), ph[3],
                                                                    (t[:, 0, 0].max() - t0) / 100.0, (t[:, :, 4].max() - t0) / 100.0))
