#!/usr/bin/env python3
"""The three stride-2 SparseConv3d layers of the CenterPoint backbone (16 -> 32, 32 -> 64, 64 -> 128): few pairs per output
row (an input feeds <= 8 outputs), so an output-stationary kernel multiplies mostly zero rows.  Time per kernel choice.
usage: stride_probe.py [iters]   (DF3D_SPLIT_KERNEL=pair selects the pair-compacted kernel)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import numpy as np
import torch
from dualfusion import ops, synth
from dualfusion.pipeline import CenterPointHotPath

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
torch.manual_seed(0)
os.environ["DF3D_EXECUTOR"] = "0"
model = CenterPointHotPath().eval().to(dev)
pts = [torch.from_numpy(synth.nusc_sweep(seed=0)).to(dev)]
with torch.no_grad():
    feats, coors = model.voxelize(pts)
    xs = model.backbone._stem(feats, coors, 1, model.grid_size_xyz)


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


for name, xin, stage in (("16->32", xs[0], "conv2"), ("32->64", xs[1], "conv3"), ("64->128", xs[2], "conv4")):
    conv = getattr(model.backbone, stage)[0]
    rb = conv._rulebook(xin)
    cin, cout = conv.in_channels, conv.out_channels
    n_in, n_out = xin.features.shape[0], rb.outids.shape[0]
    pairs = int((rb.nbr >= 0).sum())
    w = torch.randn(27, cin, cout, device=dev) * (0.7 / np.sqrt(27 * cin))
    f = torch.randn(n_in, cin, device=dev)
    line = "%-8s rows in %6d out %6d pairs %7d (%.2f per output row)" % (name, n_in, n_out, pairs, pairs / n_out)
    if ops.conv_split_supported(27, cin, cout):
        fs = ops.split_rows(f)
        packed = ops.conv_pack_weights(w)
        tiles = ops.conv_tiles(rb.nbr, cin, cout)
        us = timeit(lambda: ops.sparse_conv_split(fs, packed, rb.nbr, n_out, cin, cout, relu=True, tiles=tiles))
        line += "  split %.1f us" % us
        # rows sorted by their active-offset mask: a tile then needs only the offsets its rows use
        bits = (rb.nbr >= 0).to(torch.int32) << torch.arange(27, device=dev, dtype=torch.int32)[:, None]
        mask = bits.sum(0, dtype=torch.int32)
        order = torch.argsort(mask, stable=True).to(torch.int32)
        us_sort = timeit(lambda: torch.argsort(((rb.nbr >= 0).to(torch.int32) << torch.arange(27, device=dev, dtype=torch.int32)[:, None]).sum(0, dtype=torch.int32), stable=True))
        y0, _ = ops.sparse_conv_split(fs, packed, rb.nbr, n_out, cin, cout, relu=True)
        y1, _ = ops.sparse_conv_split(fs, packed, rb.nbr, n_out, cin, cout, relu=True, order=order)
        assert torch.equal(y0, y1)
        us = timeit(lambda: ops.sparse_conv_split(fs, packed, rb.nbr, n_out, cin, cout, relu=True, order=order))
        tm = mask.view(-1)[order.long()]
        ntile = (n_out + 127) // 128
        pad = torch.zeros(ntile * 128 - n_out, dtype=torch.int32, device=dev)
        per_tile = torch.cat([tm, pad]).view(ntile, 128)
        union = per_tile[:, 0].clone()
        for j in range(1, 128):
            union |= per_tile[:, j]
        pop = sum(((union >> b) & 1).float() for b in range(27))
        line += "  mask-sorted %.1f us (+ mask & sort %.1f us; offsets per tile %.1f of 27)" % (us, us_sort, float(pop.mean()))
    us = timeit(lambda: ops.sparse_conv_fused(f, w, rb.nbr, n_out, relu=True))
    line += "  fp32 %.1f us" % us
    print(line, flush=True)
