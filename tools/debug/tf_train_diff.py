"""Where do the reduced TransFusion training step on the GPU and its float64 composition part ways?  Forward hooks on the
stages of both detectors; prints max |a - b| / max |b| per stage (debugging aid for tests/test_gpu_tftrain.py)."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np, torch
import detgen, f64_reference as fr
from dualfusion import ops
from dualfusion.transfusion import parse_losses
ops.CONV_PRECISION = os.environ.get("PREC", "split")
DEV = torch.device("cuda:0")
B = 2
det = fr.small_transfusion_detector()
sd = detgen.det_state_dict({k: tuple(v.shape) for k, v in det.state_dict().items()})
det.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
d64 = copy.deepcopy(det).double().train()
det = det.to(DEV).train()
points, img, metas, gts, labels = fr.small_inputs(B, seed=int(os.environ.get('SEED', '0')))
points = [torch.from_numpy(p).to(DEV) for p in points]
img = torch.from_numpy(img).to(DEV)
gts = [torch.from_numpy(g) for g in gts]; labels = [torch.from_numpy(l) for l in labels]
feats, coors = det.voxelize(points)
rec = {}
def hook(tag, store):
    def f(mod, inp, out):
        def flat(o):
            if torch.is_tensor(o): return [o]
            if isinstance(o, (list, tuple)): return sum([flat(x) for x in o], [])
            if isinstance(o, dict): return sum([flat(o[k]) for k in sorted(o)], [])
            if hasattr(o, "features"): return [("sp", o)]
            return []
        store[tag] = flat(out)
    return f
names = ["pts_middle_encoder", "pts_backbone", "pts_neck", "pts_bbox_head.shared_conv", "pts_bbox_head.heatmap_head", "pts_bbox_head",
         "pts_middle_encoder.conv_input", "pts_middle_encoder.encoder_layers.encoder_layer1", "pts_middle_encoder.encoder_layers.encoder_layer2",
         "pts_middle_encoder.encoder_layers.encoder_layer3", "pts_middle_encoder.encoder_layers.encoder_layer4", "pts_middle_encoder.fusion_layer",
         "pts_middle_encoder.conv_out"]
ga, gb = {}, {}
for n in names:
    dict(det.named_modules())[n].register_forward_hook(hook(n, ga))
    dict(d64.named_modules())[n].register_forward_hook(hook(n, gb))
bga, bgb = {}, {}
def bhook(tag, store):
    def f(mod, gin, gout):
        store[tag] = [g for g in gout if g is not None]
    return f
bnames = ["pts_middle_encoder", "pts_backbone", "pts_neck", "pts_bbox_head.shared_conv", "pts_bbox_head.heatmap_head",
          "pts_bbox_head.heatmap_head.0", "pts_bbox_head.decoder.0", "pts_bbox_head.prediction_heads.0",
          "pts_bbox_head.decoder.0.multihead_attn", "pts_bbox_head.decoder.0.self_attn", "pts_bbox_head.heatmap_head.0.conv",
          "pts_bbox_head.heatmap_head.0.bn", "pts_bbox_head.heatmap_head.1", "pts_neck.deblocks.0.0", "pts_neck.deblocks.0.1",
          "pts_neck.deblocks.1.0", "pts_neck.deblocks.1.1", "pts_bbox_head.class_encoding"]
def fhook(tag, store):
    def f(mod, inp, out):
        outs = [out] if torch.is_tensor(out) else [o for o in out if torch.is_tensor(o)] if isinstance(out, (list, tuple)) else []
        for j, o in enumerate(outs):
            store.setdefault(tag + "/fwd", {})[j] = o.detach().clone()
            if o.requires_grad:
                o.register_hook(lambda g, j=j: store.setdefault(tag, {}).__setitem__(j, g.detach().clone()))
    return f
for n in bnames:
    dict(det.named_modules())[n].register_forward_hook(fhook(n, bga))
    dict(d64.named_modules())[n].register_forward_hook(fhook(n, bgb))
layer = det.pts_middle_encoder.fusion_layer
seen = {}
plain = layer.project
layer.project = lambda pts, m: seen.setdefault("p", (pts,) + tuple(plain(pts, m)))[1:]
loss, logs = det.training_step(None, [img], [dict(m) for m in metas], gts, labels, voxels=(feats, coors))
del layer.project
with fr.patched(projection=seen["p"]):
    l64 = d64.forward_train_voxels(feats.cpu().double(), coors.cpu(), B, [img.cpu().double()], [dict(m) for m in metas],
                                   [g.double() for g in gts], labels)
    loss64, logs64 = parse_losses(l64)
    loss64.backward()
def dense_of(sp):
    f, i = sp.features.detach().double().cpu(), sp.indices.long().cpu()
    vol = torch.zeros((sp.batch_size,) + tuple(sp.spatial_shape) + (f.shape[1],), dtype=torch.float64)
    vol[i[:, 0], i[:, 1], i[:, 2], i[:, 3]] = f
    return vol
for n in names:
    a, b = ga.get(n, []), gb.get(n, [])
    for j, (x, y) in enumerate(zip(a, b)):
        if isinstance(x, tuple):
            x, y = dense_of(x[1]), dense_of(y[1])
        x, y = x.detach().double().cpu(), y.detach().double().cpu()
        if x.shape != y.shape:
            print(n, j, "shape", tuple(x.shape), tuple(y.shape)); continue
        if x.numel() == 0: continue
        print("%-60s %d  err %.3e  scale %.3e  %s" % (n, j, float((x - y).abs().max() / max(1e-30, float(y.abs().max()))), float(y.abs().max()), tuple(x.shape)))
for n in bnames:
    for j in sorted(set(bga.get(n, {})) & set(bgb.get(n, {}))):
        x, y = bga[n][j].double().cpu(), bgb[n][j].double().cpu()
        print("GRAD_OUT %-50s %d  err %.3e  scale %.3e  %s" % (n, j, float((x - y).abs().max() / max(1e-30, float(y.abs().max()))), float(y.abs().max()), tuple(x.shape)))
        bad = ((x - y).abs() > 1e-4 * y.abs().max())
        if int(bad.sum()):
            fa, fb = bga[n + "/fwd"][j].double().cpu(), bgb[n + "/fwd"][j].double().cpu()
            print("      entries off by > 1e-4 of scale: %d of %d; forward values there (gpu, f64): %s" % (
                int(bad.sum()), bad.numel(), [(float(a), float(b)) for a, b in zip(fa[bad][:6], fb[bad][:6])]))
for k in logs64:
    print(k, float(logs[k]), float(logs64[k]))
want = dict(d64.named_parameters())
rows = []
l2rows = []
for k, p in det.named_parameters():
    w = want[k].grad
    if w is None or p.grad is None:
        print("nograd", k, p.grad is None, w is None); continue
    sc = float(w.abs().max())
    g = p.grad.double().cpu()
    ratio = float((g * w).sum() / max(1e-300, float((w * w).sum())))
    resid = float((g - ratio * w).abs().max()) / max(sc, 1e-30)
    e2 = float((g - w).norm() / max(1e-300, float(w.norm())))
    rows.append((float((g - w).abs().max()) / max(sc, 1e-30), k, sc, ratio, resid))
    l2rows.append(e2)
rows.sort(reverse=True)
for r in rows[:40]: print("%.3e  %-90s scale %.3e ratio %.6f resid %.2e" % r)
print("---- head params")
for r in rows:
    if r[1].startswith("pts_bbox_head"): print("%.3e  %-90s scale %.3e ratio %.6f resid %.2e" % r)
print("median", np.median([r[0] for r in rows]), "max (scale > 1e-12)", max(r[0] for r in rows if r[2] > 1e-12))
print("L2-relative: median %.3e max %.3e" % (np.median(l2rows), max(e for e, r in zip(l2rows, rows) if r[2] > 1e-12)))
