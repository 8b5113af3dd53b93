#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE calibration on a gather of KNOWN size in the conv kernel's own access pattern
(MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access pattern").

A K = 1 "convolution" 128 -> 128 over N = 2^21 rows whose neighbour table is a random permutation: every workgroup
gathers 128 distinct 512-byte split rows, every row of the 1 GiB input is read exactly once (four times the 256 MiB
Infinity Cache, so nothing is served on-die), and the output is written once as fp32 rows and once as split rows.
Expected per launch:  fetch = N * 512 (rows) + N * 4 (table) bytes,  write = 2 * N * 512 bytes.
Run under `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE`; tools/make_profiles.py turns the two
counter files into the correction factors."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

from dualfusion import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
N, C = 1 << 21, 128
f = torch.randn(N, C, device=dev)
fs = ops.split_rows(f)
del f
w = torch.randn(1, C, C, device=dev) * 0.1
packed = ops.conv_pack_weights(w)
nbr = torch.randperm(N, device=dev).to(torch.int32).view(1, N).contiguous()
for _ in range(3):
    out, osp = ops.sparse_conv_split(fs, packed, nbr, N, C, C, relu=True)
torch.cuda.synchronize()
print("calibration launches done: N=%d rows, expected fetch %d B, write %d B per launch" % (N, N * 512 + N * 4, 2 * N * 512))
