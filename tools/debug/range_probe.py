import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
from test_gpu_range import _detector, DEV
from dualfusion import ops, synth
det, sd = _detector(400.0)
points = [torch.from_numpy(synth.nusc_sweep(seed=31)).to(DEV)]
with torch.no_grad():
    preds = det._predictions(points)
    w = ops.overflow_word(reset=False).cpu()
    print("word", int(w[0]), ops.overflow_units(int(w[0]) & 0xffffffff))
    print("sync", ops.split_overflow(reset=False))
    hp = det.hot_path
    neck, hp.neck, hp.backbone.dense_layout = hp.neck, None, "nchw"
    x = hp(points)[0]
    print("bev max", float(x.abs().max()), "finite", bool(torch.isfinite(x).all()))
    print("hm max", float(preds[0]["hm"].abs().max()))
