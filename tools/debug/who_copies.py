#!/usr/bin/env python3
"""Which Python frames make the big layout copies of a training step: a TorchDispatchMode that prints the stack of every
aten::copy_ / clone / contiguous whose output has one of the shapes given on the command line (e.g. 6,40050,256)."""
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

import bench  # noqa: E402

# usage: who_copies.py [workload=cp_fusion] [stage=train] [precision=split] shape...   (shape = 6,40050,256)
words = [a for a in sys.argv[1:] if "," not in a]
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:] if "," in a] or [(6, 40050, 256)]
BIG = int(os.environ.get("DF3D_SPY_BYTES", "0"))          # also every listed operator with an output of at least so many bytes
wlname, stage, prec = (words + ["cp_fusion", "train", "split"][len(words):])[:3]
sys.argv = [sys.argv[0], "--stage", stage, "--workload", wlname, "--no-cpu-baseline", "--conv-precision", prec, "--frames", "4", "--inflight", "1"]
args = bench.parse()
dev = torch.device("cuda:0")
wl = bench.make_workload(args, 0, 1, dev)
for i in range(2):
    wl.step(i, stage)
torch.cuda.synchronize()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if isinstance(out, torch.Tensor) and (tuple(out.shape) in shapes or (BIG and out.is_cuda and out.numel() * out.element_size() >= BIG)) and any(k in name for k in ("copy_", "clone", "contiguous", "_to_copy", "addmm", "cat")):
            if True:
                fr = [f for f in traceback.format_stack() if "dualfusion" in f or "bench.py" in f]
                print("==", name, tuple(out.shape), "strides in:", [tuple(a.stride()) for a in args if isinstance(a, torch.Tensor)][:2])
                print("".join(fr[-4:]))
        return out


with Spy():
    wl.step(2, stage)
torch.cuda.synchronize()
if hasattr(wl, "close"):
    wl.close()
