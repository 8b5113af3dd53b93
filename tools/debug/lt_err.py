import copy, os, sys
sys.path[:0] = ["/root/repo", "/root/repo/3d-dual-fusion_amd", "/root/repo/tests", "/root/repo/tests/golden"]
import torch
import detgen
from dualfusion.pointformer import TransformerEncoderLayerPreNorm
dev = torch.device("cuda:0")
m = TransformerEncoderLayerPreNorm(d_model=64, nhead=4, dim_feedforward=128, dropout=0.0).eval()
sd = detgen.det_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
for G in (70, 5000):
    x = torch.from_numpy(detgen.randn("ltf_x_%d" % G, (32, G, 64))) * 1.5 + 0.3
    with torch.no_grad():
        ref = copy.deepcopy(m).double()(x.double())
        md = copy.deepcopy(m).to(dev)
        y = md(x.to(dev))
        os.environ["DF3D_LT_FUSED"] = "0"
        rows = md(x.to(dev))
        os.environ.pop("DF3D_LT_FUSED")
    scale = float(ref.abs().max())
    print("G", G, "fused vs f64 %.2e" % (float((y.cpu().double() - ref).abs().max()) / scale), "rows vs f64 %.2e" % (float((rows.cpu().double() - ref).abs().max()) / scale), "fp32 torch vs f64 %.2e" % (float((m(x).double() - ref).abs().max()) / scale))
from dualfusion import ops
print("overflow", ops.split_overflow(reset=True))
