#!/usr/bin/env python3
"""Phase times of the ping-pong conv kernel (library built with -DDF3D_OS_TRACE): ticks a wave spends in its matrix
phases, its memory phases and waiting at the barriers between them."""
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
os.environ["DF3D_OS_SK"] = "2"
model = CenterPointHotPath().eval().to(dev)
pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
with torch.no_grad():
    feats, coors = model.voxelize(pts)
    xs = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)
lib = ctypes.CDLL(_lib.LIB_PATH)
lib.df3d_debug_set_os_trace.argtypes = [ctypes.c_void_p]
for stage, x in (("conv4", xs[3]), ("conv3", xs[2])):
    blk = getattr(model.backbone, stage)[3]
    rb = x.find_indice_pair(blk.conv1.indice_key)
    C, n = x.features.shape[1], x.features.shape[0]
    w = torch.randn(27, C, C, device=dev) * 0.05
    fs = ops.split_rows(torch.randn(n, C, device=dev))
    packed = ops.conv_pack_weights(w)
    nwg = (n + 127) // 128
    tr = torch.zeros((nwg, 16, 8), dtype=torch.int64, device=dev)
    for rep in range(6):
        if rep == 5:
            lib.df3d_debug_set_os_trace(ctypes.c_void_p(tr.data_ptr()))
        ops.sparse_conv_split(fs, packed, rb.nbr, n, C, C, relu=True)
    torch.cuda.synchronize()
    lib.df3d_debug_set_os_trace(None)
    t = tr.cpu().numpy()[:, :8, :].astype(np.float64)
    turns = t[:, 0, 5]
    whole = t[:, :, 4] - t[:, :, 0]
    t0 = t[:, :, 6].min()
    print("%s: %d workgroups; ticks (10 ns) from the first wave's start: WG start mean %.0f max %.0f | loop start mean %.0f | loop end mean %.0f | "
          "WG end mean %.0f max %.0f ; per WG: prologue %.0f  loop %.0f  epilogue %.0f" % (
              stage, nwg, (t[:, :, 6] - t0).mean(), (t[:, :, 6] - t0).max(), (t[:, :, 0] - t0).mean(), (t[:, :, 4] - t0).mean(),
              (t[:, :, 7] - t0).mean(), (t[:, :, 7] - t0).max(), (t[:, :, 0] - t[:, :, 6]).mean(), whole.mean(), (t[:, :, 7] - t[:, :, 4]).mean()))
    for grp, sl in (("group 0 (waves 0-3)", slice(0, 4)), ("group 1 (waves 4-7)", slice(4, 8))):
        mma, mem, bar = t[:, sl, 1].mean(), t[:, sl, 2].mean(), t[:, sl, 3].mean()
        print("%s %s: per wave ticks mma %.0f  mem %.0f  barrier wait %.0f  | loop total %.0f (turns %.1f -> per turn mma %.0f mem %.0f bar %.0f)" % (
            stage, grp, mma, mem, bar, whole[:, sl].mean(), turns.mean(), mma / turns.mean(), mem / turns.mean(), bar / turns.mean()))
