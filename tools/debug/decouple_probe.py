"""Experimental executor decoupling (DF3D_EXEC_DECOUPLE=1): pipelined frames against isolated frames, LiDAR-only hot path
(no fusion adapter), then with the adapter.  Prints the first mismatching frame."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

from dualfusion import synth  # noqa: E402
from dualfusion.fusion import build_centerpoint_fusion, synthetic_camera_inputs  # noqa: E402
from dualfusion.pipeline import CenterPointHotPath  # noqa: E402

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "lidar"
torch.manual_seed(0)
m = CenterPointHotPath(fusion=build_centerpoint_fusion() if which == "fusion" else None).eval().to(dev)
frames = [([torch.from_numpy(synth.nusc_sweep(seed=40 + j)).to(dev)], synthetic_camera_inputs(1, dev, seed=j)) for j in range(3)]
torch.cuda.synchronize()


def run(pts, bd, ex):
    return m(pts, batch_dict=dict(bd), example=dict(ex))[0] if which == "fusion" else m(pts)[0]


with torch.no_grad():
    want = []
    for pts, (bd, ex) in frames:
        want.append(run(pts, bd, ex).clone())
        torch.cuda.synchronize()
    m.resident_inputs = True
    if m.fusion is not None:
        m.fusion.resident_inputs = True
    os.environ["DF3D_EXEC_DECOUPLE"] = "1"
    for rnd in range(6):
        got = [run(pts, bd, ex).clone() for _ in range(4) for pts, (bd, ex) in frames]
        torch.cuda.synchronize()
        bad = [k for k, y in enumerate(got) if not torch.equal(y, want[k % 3])]
        print(which, "round", rnd, "mismatching frames:", bad, flush=True)
