#!/usr/bin/env python3
"""Where the cycles of spconv_os_split_kernel go (library built with DF3D_HIPCC_FLAGS=-DDF3D_OS_TRACE): s_memtime stamps of
every wave at kernel start / after the neighbour-tile prologue / before the step loop / after it / at the end.
usage: trace_probe.py"""
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
cases = []
for stage, x in (("conv4", xs[3]), ("conv3", xs[2]), ("conv2", xs[1])):
    blk = getattr(model.backbone, stage)[3]
    rb = x.find_indice_pair(blk.conv1.indice_key)
    C = x.features.shape[1]
    cases.append((stage, rb.nbr, x.features.shape[0], C, 27))
nbr9, _, _ = ops.conv2d_neighbors(1, 180, 180, 3, 3, 1, 1, False, dev)
cases.append(("neck128", nbr9, 32400, 128, 9))
for name, nbr, n, C, K in cases:
    w = torch.randn(K, C, C, device=dev) * 0.05
    f = torch.randn(n, C, device=dev)
    fs = ops.split_rows(f)
    packed = ops.conv_pack_weights(w)
    nwg = (n + 127) // 128
    tr = torch.zeros((nwg, 16, 8), dtype=torch.int64, device=dev)
    for _ in range(3):
        ops.sparse_conv_split(fs, packed, nbr, n, C, C, relu=True)
    torch.cuda.synchronize()
    lib.df3d_debug_set_os_trace(ctypes.c_void_p(tr.data_ptr()))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    ops.sparse_conv_split(fs, packed, nbr, n, C, C, relu=True)
    b.record()
    torch.cuda.synchronize()
    lib.df3d_debug_set_os_trace(None)
    t = tr.cpu().numpy()[:, :8, :]                       # 8 waves
    t0 = t[:, :, 0].min()
    steps = t[:, 0, 5]
    d = lambda i, j: (t[:, :, j] - t[:, :, i]).astype(np.float64)
    print("%-8s rows %6d C %3d: kernel %.1f us (event) | workgroups %d, steps/WG mean %.1f max %d" % (
        name, n, C, a.elapsed_time(b) * 1e3, nwg, steps.mean(), steps.max()))
    print("   ticks (s_memtime, 100 MHz?) span first start -> last end: %d" % (t[:, :, 4].max() - t0))
    for lbl, i, j in (("prologue (nbr tile, masks)", 0, 1), ("first loads", 1, 2), ("step loop", 2, 3), ("epilogue", 3, 4), ("whole", 0, 4)):
        x = d(i, j)
        print("   %-28s mean %9.1f  min %9.1f  max %9.1f ticks" % (lbl, x.mean(), x.min(), x.max()))
    per = d(2, 3).mean(1) / np.maximum(steps, 1)
    print("   step loop / steps: mean %.2f ticks per step; start skew of workgroups (max - min start) %d ticks" % (
        per.mean(), t[:, :, 0].max() - t0))
