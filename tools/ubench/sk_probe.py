#!/usr/bin/env python3
"""spconv_os_lc_kernel (loader / consumer waves, LDS-DMA) against spconv_os_split_kernel: values and time, on one SubM layer
of every split-precision backbone stage and on the dense neck shape.  usage: sk_probe.py [iters]"""
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
    model = CenterPointHotPath().eval().to(dev)
    pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
    with torch.no_grad():
        feats, coors = model.voxelize(pts)
        xs = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)
    res = {}
    for stage, x in (("conv4", xs[3]), ("conv3", xs[2]), ("conv2", xs[1])):
        blk = getattr(model.backbone, stage)[3]
        rb = x.find_indice_pair(blk.conv1.indice_key)
        C = x.features.shape[1]
        w = torch.randn(27, C, C, device=dev) * (0.7 / np.sqrt(27 * C))
        f = torch.randn(x.features.shape[0], C, device=dev)
        fs = ops.split_rows(f)
        n = f.shape[0]
        packed = ops.conv_pack_weights(w)
        bias = torch.randn(C, device=dev)
        resid = torch.randn(n, C, device=dev)
        out, osp = ops.sparse_conv_split(fs, packed, rb.nbr, n, C, C, bias=bias, residual=resid, relu=True)
        us = timeit(lambda: ops.sparse_conv_split(fs, packed, rb.nbr, n, C, C, bias=bias, residual=resid, relu=True))
        res[stage] = (out.cpu().numpy(), osp.cpu().numpy(), us, n, C)
        print("%s C=%d rows %d: %.1f us" % (stage, C, n, us), flush=True)
    # dense neck shape
    nbr, Ho, Wo = ops.conv2d_neighbors(1, 180, 180, 3, 3, 1, 1, False, dev)
    f = torch.randn(180 * 180, 128, device=dev)
    fs = ops.split_rows(f)
    w = torch.randn(9, 128, 128, device=dev) * 0.03
    packed = ops.conv_pack_weights(w)
    out, osp = ops.sparse_conv_split(fs, packed, nbr, 32400, 128, 128, relu=True)
    us = timeit(lambda: ops.sparse_conv_split(fs, packed, nbr, 32400, 128, 128, relu=True))
    res["neck128"] = (out.cpu().numpy(), osp.cpu().numpy(), us, 32400, 128)
    print("neck 128->128 K=9 rows 32400: %.1f us" % us, flush=True)
    np.savez(sys.argv[3], **{k + "_out": v[0] for k, v in res.items()}, **{k + "_sp": v[1] for k, v in res.items()})
    sys.exit(0)

import numpy as np
iters = sys.argv[1] if len(sys.argv) > 1 else "30"
outs = {}
for tag, env in (("os", {"DF3D_OS_LC": "0"}), ("lc", {"DF3D_OS_LC": "1"})):
    print("----", tag, flush=True)
    path = "/tmp/sk_probe_%s.npz" % tag
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "child", iters, path], env=dict(os.environ, **env))
    outs[tag] = np.load(path)
for tag in list(outs)[1:]:
    for k in outs["os"].files:
        a, b = outs["os"][k], outs[tag][k]
        if k.endswith("_out"):
            print("%-10s %-14s max |diff| / max |ref| = %.3e" % (tag, k, np.abs(a - b).max() / np.abs(a).max()))
        else:
            print("%-10s %-14s split rows differing bytes: %d of %d" % (tag, k, int((a != b).sum()), a.size))
