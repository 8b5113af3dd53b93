import os, sys
sys.path[:0] = ["/root/repo", "/root/repo/3d-dual-fusion_amd"]
import torch
from dualfusion import ops
dev = torch.device("cuda:0")
for S in (150 * 267, 1000, 129):
    NI, Cin = 3, 256
    torch.manual_seed(S)
    img = torch.randn(NI, Cin, S, device=dev)
    ptrs = torch.tensor([img[i].data_ptr() for i in range(NI)], dtype=torch.int64, device=dev)
    wcat = torch.randn(144, Cin, device=dev) * 0.05
    packed = ops.imgproj_pack(wcat)
    os.environ["DF3D_IMGPROJ_DIRECT"] = "0"
    u0, g0 = ops.imgproj_split(ptrs, NI, Cin, S, packed)
    os.environ["DF3D_IMGPROJ_DIRECT"] = "1"
    u1, g1 = ops.imgproj_split(ptrs, NI, Cin, S, packed)
    torch.cuda.synchronize()
    print(S, "rows equal", torch.equal(u0, u1), "gate equal", torch.equal(g0, g1))
