#!/usr/bin/env python3
"""F.linear (hipBLASLt fp32) against the split-precision conv kernel over an identity table (K = 1) at the training step's
tall-skinny shapes, with the operand split and the per-step filter pack included."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

from dualfusion import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, iters=10):
    for _ in range(2):
        y = fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(iters):
        y = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / iters, y


for n, cin, cout in ((32034, 128, 1024), (32034, 256, 128), (32034, 512, 128), (240300, 128, 128), (240300, 256, 128), (32034, 128, 128)):
    x = torch.randn(n, cin, device=dev)
    w = torch.randn(cout, cin, device=dev) * 0.05
    b = torch.randn(cout, device=dev)
    t0, ref = timeit(lambda: torch.nn.functional.linear(x, w, b))
    if not ops.conv_split_supported(1, cin, cout if cout <= 128 else 128):
        print(n, cin, cout, "unsupported")
        continue
    tab = ops.identity_table(n, dev)

    def split_path():
        wt = w.t().contiguous().view(1, cin, cout)
        if cout > 128:
            blocks = wt.view(1, cin, cout // 128, 128).permute(2, 0, 1, 3).contiguous()
            out, _ = ops.conv_rows_split(ops.split_rows(x), cin, 0, ops.conv_pack_weights_groups(blocks), 128, cout // 128, tab, n, b)
            return out
        out, _ = ops.sparse_conv_split(ops.split_rows(x), ops.conv_pack_weights(wt), tab, n, cin, cout, bias=b, emit_split=False)
        return out
    t1, got = timeit(split_path)
    print("%d rows %d -> %d: F.linear %.0f us, split kernel (split + pack + conv) %.0f us, max diff %.1e of scale"
          % (n, cin, cout, t0, t1, float((got - ref).abs().max() / ref.abs().max())), flush=True)
