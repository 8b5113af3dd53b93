"""Host-side cost of bench.py's detector step, section by section, and what a SINGLE host thread achieves with two frames
in flight (frame k on stream k % 2, detector replica k % 2).
  (a) every step starts with the GPU idle (synchronize first): the time until step() returns is interpreter + C-ABI launch
      cost + the short side-stream round trips (voxel count, strided output counts, longest camera list), per section;
  (b) K steps back to back on one stream (bench.py's timed loop);
  (c) K steps alternating over two streams / replicas from the same thread.
usage: host_enqueue.py [workload] [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch  # noqa: E402

import bench  # noqa: E402


class A(object):
    workload, frames, batch, inflight = sys.argv[1] if len(sys.argv) > 1 else "cp_fusion", 8, 0, 1
    prefetch = os.environ.get("DF3D_PROBE_PREFETCH", "1") == "1"


steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
wl = bench.make_workload(A(), 0, 1, dev)
for k in range(12):
    wl.step(k, "detect")
torch.cuda.synchronize()

# ---- (a) sections, GPU idle at the start of every step
acc = {}


def wrap(obj, name, label):
    fn = getattr(obj, name)

    def timed(*a, **kw):
        t0 = time.perf_counter()
        try:
            return fn(*a, **kw)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    setattr(obj, name, timed)
    return fn


m = wl.model
hp = m.hot_path
saved = [(hp, "voxelize", wrap(hp, "voxelize", "voxelize (incl. count round trip)")),
         (hp.backbone, "_stem", wrap(hp.backbone, "_stem", "backbone stem (executor, incl. 3 count round trips)")),
         (hp.backbone, "_tail", wrap(hp.backbone, "_tail", "backbone tail (extra conv + dense rows)")),
         (m.bbox_head, "forward", wrap(m.bbox_head, "forward", "head forward")),
         (m.bbox_head, "loss_device", wrap(m.bbox_head, "loss_device", "loss_device"))]
if hp.fusion is not None:
    saved.append((hp.fusion, "forward", wrap(hp.fusion, "forward", "fusion adapter + ACTR")))
if hp.neck is not None:
    saved.append((hp.neck, "forward_rows", wrap(hp.neck, "forward_rows", "neck")))
N = 24
tot = 0.0
for k in range(N):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wl.step(k, "detect")
    tot += time.perf_counter() - t0
torch.cuda.synchronize()
print("(a) host time per step with the GPU idle at its start: %.3f ms" % (tot / N * 1e3))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("      %-58s %.3f ms" % (k, v / N * 1e3))
print("      %-58s %.3f ms" % ("(rest: bench.py glue, fresh inputs)", (tot - sum(acc.values())) / N * 1e3))
for obj, name, fn in saved:
    try:
        delattr(obj, name)
    except AttributeError:
        setattr(obj, name, fn)

def ahead_stats(w, reset=False):
    a = getattr(w.model.hot_path, "_ahead", None)
    if a is None:
        return "no worker thread"
    st = a.stats
    msg = "native worker: submit %.3f ms, take %.3f ms of which waiting for the worker %.3f ms (per frame, host)" % (
        st["submit_s"] / max(st["submits"], 1) * 1e3, st["take_s"] / max(st["takes"], 1) * 1e3,
        st["take_wait_s"] / max(st["takes"], 1) * 1e3)
    if reset:
        for k in st:
            st[k] = 0 if isinstance(st[k], int) else 0.0
    return msg


print("      " + ahead_stats(wl, reset=True))
# ---- (b) back to back, one stream
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(steps):
    wl.step(k, "detect")
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
el = time.perf_counter() - t0
print("(b) one stream: %.3f ms per step (host returned after %.3f ms per step) = %.1f sweeps/s" % (
    el / steps * 1e3, t_enq / steps * 1e3, steps / el))
print("      " + ahead_stats(wl, reset=True))
if os.environ.get("DF3D_PROBE_TRACE"):
    sys.exit(0)

# ---- (c) two streams / replicas, one thread
F = int(os.environ.get("DF3D_PROBE_INFLIGHT", "2"))
wls = [wl] + [bench.make_workload(A(), 0, 1, dev) for _ in range(F - 1)]
for w in wls:
    w.stride = F
streams = [torch.cuda.Stream() for _ in wls]
for k in range(8 * F):
    with torch.cuda.stream(streams[k % F]):
        wls[k % F].step(k, "detect")
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    for k in range(steps):
        with torch.cuda.stream(streams[k % F]):
            wls[k % F].step(k, "detect")
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print("(c) %d streams / replicas, one thread: %.3f ms per step (host returned after %.3f) = %.1f sweeps/s" % (
        F, el / steps * 1e3, t_enq / steps * 1e3, steps / el))
    print("      " + ahead_stats(wls[0], reset=True))
for w in wls:
    w.close()
