#!/usr/bin/env python3
"""lc_trace.py for the head's middle convolutions (64 -> 36 x 64, 3 x 3 on the 180 x 180 map = 18 column blocks of 128):
where the waves of the loader / consumer kernel spend their time (library built with -DDF3D_OS_TRACE)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import numpy as np
import torch
from dualfusion import _lib, ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
H = W = 180
G, cin, cout = 18, 64, 128
nbr, _, _ = ops.conv2d_neighbors(1, H, W, 3, 3, 1, 1, False, dev)
n = nbr.shape[1]
x = ops.split_rows(torch.randn(n, cin, device=dev))
packed = torch.cat([ops.conv_pack_weights(torch.randn(9, cin, cout, device=dev) * 0.05) for _ in range(G)])
sc = torch.ones(G * cout, device=dev)
lib = ctypes.CDLL(_lib.LIB_PATH)
traced = hasattr(lib, "df3d_debug_set_os_trace")     # only in builds with -DDF3D_OS_TRACE; the launch time is printed either way
if traced:
    lib.df3d_debug_set_os_trace.argtypes = [ctypes.c_void_p]
nwg = (n + 127) // 128 * G
tr = torch.zeros((nwg, 16, 8), dtype=torch.int64, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(6):
    if rep == 5 and traced:
        lib.df3d_debug_set_os_trace(ctypes.c_void_p(tr.data_ptr()))
    if rep == 4:
        e0.record()
    ops.conv_rows_split(x, cin, 0, packed, cout, G, nbr, n, None, sc, sc, relu=True, want_out=False, want_split=True)
    if rep == 4:
        e1.record()
torch.cuda.synchronize()
print("launch %.1f us (DF3D_LC_CBW=%s)" % (e0.elapsed_time(e1) * 1e3, os.environ.get("DF3D_LC_CBW", "rule")))
if not traced:
    sys.exit(0)
lib.df3d_debug_set_os_trace(None)
t = tr.cpu().numpy().astype(np.float64)
steps = t[:, 0, 5].mean()
print("%d workgroups, %.1f steps per tile; ticks (10 ns): prologue %.0f  loop %.0f  epilogue %.0f" % (
    nwg, steps, (t[:, :4, 0] - t[:, :4, 6]).mean(), (t[:, :4, 4] - t[:, :4, 0]).mean(), (t[:, :4, 7] - t[:, :4, 4]).mean()))
m = t[:, 0:4, :]
print("matrix waves: per step  work %.0f  barrier %.0f" % (m[:, :, 1].mean() / steps, m[:, :, 3].mean() / steps))
l = t[:, 4:12, :]
print("loader waves: per step  issue %.0f  landing wait %.0f  barrier %.0f" % (l[:, :, 1].mean() / steps, l[:, :, 2].mean() / steps,
                                                                               l[:, :, 3].mean() / steps))
span = t[:, :4, 7].max() - t[:, :4, 6].min()
print("first start to last end: %.0f ticks; workgroup lifetime %.0f ticks" % (span, (t[:, :4, 7] - t[:, :4, 6]).mean()))
