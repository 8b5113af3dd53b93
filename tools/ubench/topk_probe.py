#!/usr/bin/env python3
"""df3d_topk_keys on CenterHead-like keys (6 segments x 32 400 pixels, scores in [0.1, 0.12], half masked, k = 1000) and
TransFusionHead-like keys (1 x 324 000, k = 200): time per call and the region counters of the select."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "3d-dual-fusion_amd"))
from dualfusion import _lib, ops  # noqa: E402

dev = "cuda:0"
rs = np.random.RandomState(0)
for name, S, n, k, lo, hi, masked in (("centerhead", 6, 32400, 1000, 0.1, 0.12, 0.5), ("transfusion", 1, 324000, 200, 0.0, 1.0, 0.0)):
    score = rs.uniform(lo, hi, size=(S, n)).astype(np.float32)
    inv = np.uint64(0x3F800000) - score.view(np.uint32).astype(np.uint64)
    keys = (np.arange(S, dtype=np.uint64)[:, None] << np.uint64(56)) | (inv << np.uint64(24)) | np.arange(n, dtype=np.uint64)[None]
    keys = np.where(rs.uniform(size=(S, n)) < masked, np.uint64(0xFFFFFFFFFFFFFFFF), keys)
    kd = torch.from_numpy(keys.view(np.int64)).to(dev)
    lib = _lib.load()
    nbytes = lib.df3d_topk_keys_workspace_bytes(S, n, k)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=dev)
    out = torch.empty((S, k), dtype=torch.int64, device=dev)
    cnt = torch.empty((S,), dtype=torch.int32, device=dev)

    def call():
        rc = lib.df3d_topk_keys(kd.data_ptr(), S, n, k, out.data_ptr(), cnt.data_ptr(), ws.data_ptr(), nbytes, None)
        assert rc == 0
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        call()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / 100 * 1e6
    a = (S * 4096 * 4 + 255) // 256 * 256
    off = 2 * a
    regs = ws[off:off + S * 8].cpu().numpy().view(np.uint32).reshape(S, 2)
    ok = np.array_equal(out.cpu().numpy().view(np.uint64), np.sort(keys, axis=1)[:, :k])
    print("%s: %.1f us per call, regions (below threshold, ties) %s, equals full sort: %s" % (name, us, regs.tolist(), ok))
