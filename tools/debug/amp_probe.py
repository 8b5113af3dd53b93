import sys, os
sys.path.insert(0, "/root/repo/3d-dual-fusion_amd"); sys.path.insert(0, "/root/repo/tests")
import torch, numpy as np
from dualfusion import ops, synth
from dualfusion.pipeline import NUSC_TASKS, CenterPointDetector
DEV = torch.device("cuda:0")
pts = [torch.from_numpy(synth.nusc_sweep(seed=12)).to(DEV)]
tg = synth.centerhead_targets(1, [t["num_class"] for t in NUSC_TASKS], seed=5)
ex = {k: [torch.from_numpy(a).to(DEV) for a in v] for k, v in tg.items()}
torch.manual_seed(1)
det = CenterPointDetector().to(DEV).train()
params = {n: p for n, p in det.named_parameters() if p.dim() >= 4}
state = {k: v.detach().clone() for k, v in det.state_dict().items()}
res = {}
for mode in ("split", "fp32", "bf16"):
    ops.CONV_PRECISION = mode
    det.load_state_dict(state)
    det.zero_grad(set_to_none=True)
    rets = det.training_step(pts, {k: list(v) for k, v in ex.items()})
    res[mode] = ([float(v) for v in rets["loss"]], {k: p.grad.detach().double().clone() for k, p in params.items() if p.grad is not None})
print(res["split"][0]); print(res["bf16"][0])
for k in list(params)[:60]:
    if k not in res["split"][1]: continue
    b = res["split"][1][k].flatten()
    out = []
    for m in ("fp32", "bf16"):
        a = res[m][1][k].flatten()
        out.append("%s cos %.5f norm %.3f" % (m, float((a @ b) / (a.norm() * b.norm())), float(a.norm() / b.norm())))
    print("%-50s |g| %.3e  %s" % (k[-50:], float(b.norm()), "  ".join(out)))
