#!/usr/bin/env python3
"""Experiment: neck + head of the CenterPoint detector (fixed shapes, ~20 launches) re-issued from the launch tape on a stream
against the same launches replayed from a hipGraph captured over the tape: what a dependent launch costs on the GPU side."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402
from dualfusion import synth  # noqa: E402
from dualfusion.pipeline import CenterPointDetector  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
det = CenterPointDetector().eval().to(dev)
det.launch_tape = True
pts = [torch.from_numpy(synth.nusc_sweep(seed=1)).to(dev)]
hp = det.hot_path
with torch.no_grad():
    hp.defer_neck = True
    bev, _ = hp(pts)
    hp.defer_neck = False
    for _ in range(4):
        preds = det._taped_tail(bev)
    torch.cuda.synchronize()
    print("tape stats", det._tail_tape.stats)

    def timeit(fn, n=50):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) * 1e3 / n
    print("tape replay on the stream: %.1f us" % timeit(lambda: det._taped_tail(bev)))
    want = [v.clone() for d in preds for _, v in sorted(d.items())]
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            out = det._taped_tail(bev)
    torch.cuda.synchronize()
    print("hipGraph over the tape:    %.1f us" % timeit(g.replay))
    got = [v.clone() for d in out for _, v in sorted(d.items())]
    print("same maps:", all(torch.equal(a, b) for a, b in zip(got, want)), "launches:", len(det._tail_tape.tapes[next(iter(det._tail_tape.tapes))].calls))
