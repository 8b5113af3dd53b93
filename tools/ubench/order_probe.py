#!/usr/bin/env python3
"""Tiling order of the output-stationary conv kernel: time one SubM layer of every backbone stage under candidate
row orders (computed on the host here): flat (storage order), BEV blocks of 8..64 cells in Morton order [+ neighbour
mask inside a block], mask alone.  usage: order_probe.py [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dualfusion import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
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


def morton(y, x):
    r = np.zeros_like(y, dtype=np.int64)
    for b in range(12):
        r |= ((x >> b) & 1) << (2 * b)
        r |= ((y >> b) & 1) << (2 * b + 1)
    return r


os.environ["DF3D_EXECUTOR"] = "0"
from dualfusion import synth  # noqa: E402
from dualfusion.pipeline import CenterPointHotPath  # noqa: E402
model = CenterPointHotPath().eval().to(dev)
pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
with torch.no_grad():
    feats, coors = model.voxelize(pts)
    xs = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)
for stage, x in (("conv4", xs[3]), ("conv3", xs[2]), ("conv2", xs[1])):
    blk = getattr(model.backbone, stage)[3]
    rb = x.find_indice_pair(blk.conv1.indice_key)
    C = x.features.shape[1]
    w = blk.conv1.weight.detach().view(-1, C, C).contiguous()
    fs = ops.split_rows(x.features.contiguous())
    n = x.features.shape[0]
    nbr = rb.nbr
    valid = (nbr >= 0).cpu().numpy()
    ind = x.indices.cpu().numpy().astype(np.int64)
    mask = (valid * (1 << np.arange(27))[:, None]).sum(0)
    y_, x_ = ind[:, 2], ind[:, 3]
    orders = {"flat": None, "mask": np.argsort(mask, kind="stable")}
    for bs in (3, 4, 5, 6):
        blkid = morton(y_ >> bs, x_ >> bs)
        orders["blk%d,flat" % (1 << bs)] = np.lexsort((np.arange(n), blkid))
        orders["blk%d,mask" % (1 << bs)] = np.lexsort((mask, blkid))
        orders["blk%d,z,mask" % (1 << bs)] = np.lexsort((mask, ind[:, 1], blkid))
    packed = ops.conv_pack_weights(w)
    ref_out = None
    for name, o in orders.items():
        ot = None if o is None else torch.from_numpy(o.astype(np.int32)).to(dev)
        out, _ = ops.sparse_conv_split(fs, packed, nbr, n, C, C, relu=True, order=ot)
        if ref_out is None:
            ref_out = out.clone()
        ok = bool(torch.equal(out, ref_out))
        v2 = valid if o is None else valid[:, o]
        T = 128
        nt = (n + T - 1) // T
        pad = np.zeros((27, nt * T), bool)
        pad[:, :n] = v2
        act = pad.reshape(27, nt, T).any(2).mean()
        us = timeit(lambda: ops.sparse_conv_split(fs, packed, nbr, n, C, C, relu=True, order=ot))
        print("%-6s C=%3d rows %6d  %-14s active(T128) %.3f : %7.1f us  %s" % (stage, C, n, name, act, us,
                                                                               "" if ok else "MISMATCH"))
