#!/usr/bin/env python3
"""Ablation of spconv_os_split_kernel on conv4 / conv3 / the dense neck layer under DF3D_OS_DBG bits
(needs a library built with DF3D_HIPCC_FLAGS=-DDF3D_OS_EXPERIMENTS): 1 no gathers, 2 no W staging, 4 no MFMAs,
16 no B-fragment LDS reads, 32 no output stores.  usage: ablate_probe.py [iters]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np
    import torch
    from dualfusion import ops, synth
    from dualfusion.pipeline import CenterPointHotPath
    iters = int(sys.argv[2])
    dev = torch.device("cuda:0")
    torch.manual_seed(0)

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / iters
    os.environ["DF3D_EXECUTOR"] = "0"
    dbg = os.environ.get("DF3D_OS_DBG", "0")
    os.environ["DF3D_OS_DBG"] = "0"
    model = CenterPointHotPath().eval().to(dev)
    pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
    with torch.no_grad():
        feats, coors = model.voxelize(pts)
        xs = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)
    os.environ["DF3D_OS_DBG"] = dbg
    line = []
    for stage, x in (("conv4", xs[3]), ("conv3", xs[2])):
        blk = getattr(model.backbone, stage)[3]
        rb = x.find_indice_pair(blk.conv1.indice_key)
        C = x.features.shape[1]
        w = torch.randn(27, C, C, device=dev) * (0.7 / np.sqrt(27 * C))
        f = torch.randn(x.features.shape[0], C, device=dev)
        fs = ops.split_rows(f)
        n = f.shape[0]
        packed = ops.conv_pack_weights(w)
        us = timeit(lambda: ops.sparse_conv_split(fs, packed, rb.nbr, n, C, C, relu=True))
        line.append("%s %6.1f" % (stage, us))
    nbr, Ho, Wo = ops.conv2d_neighbors(1, 180, 180, 3, 3, 1, 1, False, dev)
    f = torch.randn(180 * 180, 128, device=dev)
    fs = ops.split_rows(f)
    packed = ops.conv_pack_weights(torch.randn(9, 128, 128, device=dev) * 0.03)
    us = timeit(lambda: ops.sparse_conv_split(fs, packed, nbr, 32400, 128, 128, relu=True))
    line.append("neck128 %6.1f" % us)
    print("DBG=%-3s %s" % (dbg, " | ".join(line)), flush=True)
    sys.exit(0)

iters = sys.argv[1] if len(sys.argv) > 1 else "30"
for dbg in [int(x) for x in os.environ.get("ABLATE_SET", "0,1,2,16,4,32,3,19,23,55,20,5").split(",")]:
    subprocess.call([sys.executable, os.path.abspath(__file__), "child", iters], env=dict(os.environ, DF3D_OS_DBG=str(dbg)))
