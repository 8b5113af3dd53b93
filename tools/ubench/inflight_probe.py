#!/usr/bin/env python3
"""Throughput of the configs[1] hot path with 1, 2 or 3 frames in flight (one host thread + HIP stream + model replica
per frame slot; frames are independent).  usage: inflight_probe.py [steps]"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
torch.cuda.set_device(0)


def slot(i):
    model = bench.build_model("cp_fusion", dev)
    pts, extra = bench.make_inputs("cp_fusion", 1, i, dev)
    return model, pts, extra, torch.cuda.Stream()


for nslots in (1, 2, 3):
    slots = [slot(i) for i in range(nslots)]

    def work(s, n):
        model, pts, extra, stream = s
        with torch.cuda.stream(stream):
            for _ in range(n):
                bench.run_step(model, pts, extra)

    for s in slots:
        work(s, 3)
    torch.cuda.synchronize()
    per = steps // nslots
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(s, per)) for s in slots]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("frames in flight %d: %d sweeps in %.1f ms -> %.1f sweeps/s" % (nslots, per * nslots, el * 1e3, per * nslots / el))
