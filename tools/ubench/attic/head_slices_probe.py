#!/usr/bin/env python3
"""CenterHead.forward_rows on a 512-channel 180 x 180 map with the branches' two last depths in 1 / 2 / 3 / 6 / 9 / 18 slices
(DF3D_HEAD_SLICES): a slice's [pixels, 64 x branches] activation stays in the Infinity Cache between the launch that writes it
and the launch that reads it.  Values must be identical; time per forward."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

from dualfusion.heads import CenterHead  # noqa: E402
from dualfusion.pipeline import NUSC_CODE_WEIGHTS, NUSC_COMMON_HEADS, NUSC_TASKS  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
head = CenterHead(in_channels=512, tasks=NUSC_TASKS, dataset='nuscenes', weight=0.25, code_weights=NUSC_CODE_WEIGHTS,
                  common_heads=dict(NUSC_COMMON_HEADS), share_conv_channel=64, dcn_head=False).eval().to(dev)
x = torch.randn(1, 180, 180, 512, device=dev).permute(0, 3, 1, 2) * 0.5
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ref = None
for rnd in range(2):
    for ns in (1, 2, 3, 6, 9, 18):
        os.environ["DF3D_HEAD_SLICES"] = str(ns)
        with torch.no_grad():
            for _ in range(3):
                out = head(x)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            for _ in range(iters):
                out = head(x)
            b.record()
            torch.cuda.synchronize()
        flat = torch.cat([out[t][k].reshape(-1) for t in range(len(out)) for k in sorted(out[t])])
        if ref is None:
            ref = flat.clone()
        print("slices %2d: %.1f us per head forward, identical to 1 slice: %s" % (ns, a.elapsed_time(b) * 1e3 / iters,
                                                                                torch.equal(flat, ref)), flush=True)
