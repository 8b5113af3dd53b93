#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz FROM THE REFERENCE ITSELF.

Runs only where /root/reference exists (this container).  Sources of truth:
  * oracle/_ref/voxel_layer.so, sparse_conv_ext.so  = the reference's own C++ CPU code
    (TF/mmdet3d/ops/voxel/src/voxelization_cpu.cpp, TF/mmdet3d/ops/spconv/src/*.cc),
    compiled by oracle/build_ref.py;
  * the reference's Python modules imported from /root/reference/CenterPoint/det3d with
    dependency stubs in sys.modules (cv2 / torchvision / mmcv / the CUDA-only extension
    modules are absent here; the stubs only satisfy `import`, no reference arithmetic is
    replaced except MSDeformAttnFunction.apply -> the reference's OWN pure-torch
    ms_deform_attn_core_pytorch, exactly what the reference's ops/test.py compares the
    CUDA kernel with);
  * literal expected values copied as DATA from the reference's tests
    (TF/tests/test_models/test_voxel_encoder/test_voxel_generator.py:15-22).
The fixtures are data: inputs (or the seeds that regenerate them, tests/golden/detgen.py)
and expected outputs.  No reference source text is stored.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "3d-dual-fusion_amd"))
sys.path.insert(0, HERE)

import detgen  # noqa: E402
from oracle import ref  # noqa: E402
from dualfusion import synth  # noqa: E402


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print("wrote %s (%.1f KB)" % (name, os.path.getsize(path) / 1024))


# ------------------------------------------------------------------ voxelize
def gen_voxelize():
    out = {}
    # (a) the reference's own golden vector, test_voxel_generator.py:6-22
    np.random.seed(0)
    pts = np.random.rand(1000, 4).astype(np.float32)
    v, c, n = ref.hard_voxelize(pts, [0.5, 0.5, 0.5], [0, -40, -3, 70.4, 40, 1], 1000, 20000)
    out["tg_points"] = pts
    out["tg_expected_coors"] = np.array([[7, 81, 1], [6, 81, 0], [7, 80, 1], [6, 81, 1],
                                         [7, 81, 0], [6, 80, 1], [7, 80, 0], [6, 80, 0]], np.int32)
    out["tg_expected_num"] = np.array([120, 121, 127, 134, 115, 127, 125, 131], np.int32)
    out["tg_ref_coors"], out["tg_ref_num"] = c, n
    out["tg_ref_voxel_sum"] = v.sum(1)
    # (b) nuScenes-shaped sweep slice, no cap / cap hit / max_points hit
    sw = synth.nusc_sweep(seed=7)[:6000]
    out["sw_points"] = sw
    for tag, maxp, maxv in (("nocap", 10, 20000), ("cap", 10, 3000), ("mp3", 3, 20000)):
        v, c, n = ref.hard_voxelize(sw, synth.NUSC_VOXEL, synth.NUSC_RANGE, maxp, maxv)
        out["sw_%s_coors" % tag], out["sw_%s_num" % tag] = c, n
        out["sw_%s_voxel_sum" % tag] = v.sum(1)
        out["sw_%s_first" % tag] = v[:, 0].copy()
        out["sw_%s_params" % tag] = np.array([maxp, maxv], np.int64)
    save("voxelize.npz", **out)


# ------------------------------------------------------------------ rulebook + conv
RB_CASES = {
    # name: (ksize, stride, padding, dilation, subm, cin, cout)
    "subm3": ([3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], 1, 16, 16),
    "conv_s2p1": ([3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], 0, 16, 32),
    "conv_s2p011": ([3, 3, 3], [2, 2, 2], [0, 1, 1], [1, 1, 1], 0, 8, 16),
    "conv_k311": ([3, 1, 1], [2, 1, 1], [0, 0, 0], [1, 1, 1], 0, 16, 16),
    "subm3_c5": ([3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], 1, 5, 16),
}
RB_SHAPE = [11, 40, 36]
RB_BATCH = 2


def rb_indices():
    return detgen.clustered_voxels("rb", RB_BATCH, RB_SHAPE, n_seeds=6, walk=260)


def gen_rulebook_conv():
    ind = rb_indices()
    out = {"indices": ind, "shape": np.array(RB_SHAPE), "batch": np.array(RB_BATCH)}
    for name, (ks, st, pd, dl, subm, cin, cout) in RB_CASES.items():
        outids, pairs, num, oshape = ref.get_indice_pairs(ind, RB_BATCH, RB_SHAPE, ks, st, pd, dl, subm)
        feats = detgen.randn("feat_" + name, (len(ind), cin))
        filt = detgen.randn("filt_" + name, (ks[0], ks[1], ks[2], cin, cout), 0.2)
        y = ref.indice_conv(feats, filt, pairs, num, len(outids), subm)
        out[name + "_outids"] = outids
        out[name + "_pairs"] = pairs
        out[name + "_num"] = num
        out[name + "_oshape"] = np.array(oshape)
        out[name + "_y"] = y
    save("rulebook_conv.npz", **out)


# ------------------------------------------------------------------ reference python import
def _stub(n, **a):
    m = types.ModuleType(n)
    m.__dict__.update(a)
    sys.modules[n] = m
    return m


def import_reference_actr():
    R = "/root/reference/CenterPoint/det3d"
    for pkg, path in [("det3d", R), ("det3d.models", R + "/models"),
                      ("det3d.models.model_utils", R + "/models/model_utils"),
                      ("det3d.models.model_utils.ops", R + "/models/model_utils/ops"), ("det3d.ops", R + "/ops")]:
        _stub(pkg).__path__ = [path]
    _stub("cv2")
    tv = _stub("torchvision", __version__="0.25.0")
    tv.ops = _stub("torchvision.ops")
    tv.ops.misc = _stub("torchvision.ops.misc", _NewEmptyTensorOp=None)
    _stub("MultiScaleDeformableAttention")

    class ConvModule(torch.nn.Module):
        def __init__(s, *a, **k):
            super().__init__()

    _stub("mmcv")
    _stub("mmcv.cnn", ConvModule=ConvModule)
    for n, a in [("det3d.ops.gather_points.gather_points", "gather_points"),
                 ("det3d.ops.furthest_point_sample.points_sampler", "Points_Sampler"),
                 ("det3d.ops.group_points.group_points", "QueryAndGroup")]:
        _stub(n.rsplit(".", 1)[0])
        _stub(n, **{a: None})
    actr = importlib.import_module("det3d.models.model_utils.actr")
    func = importlib.import_module("det3d.models.model_utils.ops.functions.ms_deform_attn_func")

    class _F:
        apply = staticmethod(lambda v, s, l, loc, w, step: func.ms_deform_attn_core_pytorch(v, s, loc, w))

    importlib.import_module("det3d.models.model_utils.ops.modules.ms_deform_attn").MSDeformAttnFunction = _F
    return actr, func


class Cfg(dict):
    __getattr__ = dict.__getitem__


def gen_msda(func):
    out = {}
    # (a) the reference's own self-check vector, ops/test.py:21-46 (manual_seed(3), fp32 leg)
    N, M, D = 1, 2, 2
    Lq, L, P = 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = int(sum(h * w for h, w in shapes.tolist()))
    torch.manual_seed(3)
    value = torch.rand(N, S, M, D) * 0.01
    loc = torch.rand(N, Lq, M, L, P, 2)
    aw = torch.rand(N, Lq, M, L, P) + 1e-5
    aw /= aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
    y = func.ms_deform_attn_core_pytorch(value, shapes, loc, aw)
    out.update(t_value=value.numpy(), t_shapes=shapes.numpy(), t_loc=loc.numpy(), t_aw=aw.numpy(), t_out=y.numpy())
    # (b) hot shape: D=16, M=8, L=1, P=4; locations spill over the border
    N, M, D, Lq, L, P, H, W = 2, 8, 16, 300, 1, 4, 30, 52
    value = torch.from_numpy(detgen.randn("msda_h_value", (N, H * W, M, D)))
    loc = torch.from_numpy(detgen.rand("msda_h_loc", (N, Lq, M, L, P, 2), -0.15, 1.15))
    aw = torch.softmax(torch.from_numpy(detgen.randn("msda_h_aw", (N, Lq, M, L * P))), -1).view(N, Lq, M, L, P)
    y = func.ms_deform_attn_core_pytorch(value, torch.as_tensor([(H, W)]), loc, aw)
    out.update(h_dims=np.array([N, M, D, Lq, L, P, H, W]), h_out=y.numpy())
    # (c) multi-level, odd head dim
    N, M, D, Lq, L, P = 3, 4, 6, 50, 3, 2
    shp = [(9, 13), (5, 7), (3, 4)]
    S = sum(h * w for h, w in shp)
    value = torch.from_numpy(detgen.randn("msda_m_value", (N, S, M, D)))
    loc = torch.from_numpy(detgen.rand("msda_m_loc", (N, Lq, M, L, P, 2), -0.1, 1.1))
    aw = torch.softmax(torch.from_numpy(detgen.randn("msda_m_aw", (N, Lq, M, L * P))), -1).view(N, Lq, M, L, P)
    y = func.ms_deform_attn_core_pytorch(value, torch.as_tensor(shp), loc, aw)
    out.update(m_dims=np.array([N, M, D, Lq, L, P]), m_shapes=np.array(shp), m_out=y.numpy())
    save("msda.npz", **out)


MSDA_BWD_CASES = dict(hot=dict(N=1, M=8, D=16, Lq=40, P=4, shapes=[(8, 10)]),
                      multi=dict(N=2, M=4, D=6, Lq=20, P=2, shapes=[(5, 7), (3, 4), (2, 3)]))


def msda_bwd_inputs(tag):
    c = MSDA_BWD_CASES[tag]
    N, M, D, Lq, P, shp = c["N"], c["M"], c["D"], c["Lq"], c["P"], c["shapes"]
    L, S = len(shp), sum(h * w for h, w in shp)
    value = detgen.randn("msdab_%s_value" % tag, (N, S, M, D))
    loc = detgen.rand("msdab_%s_loc" % tag, (N, Lq, M, L, P, 2), -0.15, 1.15)
    aw = detgen.rand("msdab_%s_aw" % tag, (N, Lq, M, L, P), 0.05, 1.0)
    gout = detgen.randn("msdab_%s_gout" % tag, (N, Lq, M * D))
    return value, shp, loc, aw, gout


def gen_msda_bwd(func):
    """Gradients of the reference's own pure-torch core (ms_deform_attn_func.py:41-61) by autograd in float64 --
    what the reference's ops/test.py:49-86 checks its CUDA backward against."""
    out = {}
    for tag in MSDA_BWD_CASES:
        value, shp, loc, aw, gout = msda_bwd_inputs(tag)
        v = torch.from_numpy(value).double().requires_grad_(True)
        lo = torch.from_numpy(loc).double().requires_grad_(True)
        a = torch.from_numpy(aw).double().requires_grad_(True)
        y = func.ms_deform_attn_core_pytorch(v, torch.as_tensor(shp), lo, a)
        y.backward(torch.from_numpy(gout).double())
        out.update({tag + "_gv": v.grad.float().numpy(), tag + "_gl": lo.grad.float().numpy(),
                    tag + "_ga": a.grad.float().numpy()})
    save("msda_bwd.npz", **out)


ACTR_CFG = dict(fusion_method="sum", feature_modal="hybrid",
                hybrid_cfg=dict(attn_layer="BiGateSum1D_2", q_method="sum", q_rep_place=["weight"]),
                num_bins=80, num_channels=[256], query_num_feat=128, num_enc_layers=2,
                max_num_ne_voxel=26000, pos_encode_method="depth")
ACTR_DIMS = dict(n=2, q=150, h=12, w=20)


def actr_inputs():
    d = ACTR_DIMS
    n, q, h, w = d["n"], d["q"], d["h"], d["w"]
    v_feat = detgen.randn("actr_v_feat", (n, q, 128))
    grid = detgen.rand("actr_grid", (n, q, 2))
    i_feat = detgen.randn("actr_i_feat", (n, 256, h, w))
    lidar_grid = detgen.rand("actr_lidar", (n, q, 3), -50, 50)
    v_i_feat = detgen.randn("actr_v_i_feat", (n, q, 256))
    # zero-padded tail rows, like the per-camera padded batches the adapter builds
    for a in (v_feat, grid, lidar_grid, v_i_feat):
        a[1, q - 30:] = 0
    return v_feat, grid, i_feat, lidar_grid, v_i_feat


def gen_actr(actr):
    model = actr.build(Cfg(ACTR_CFG), model_name="ACTR").eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = detgen.det_state_dict(shapes)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    v_feat, grid, i_feat, lidar_grid, v_i_feat = [torch.from_numpy(a) for a in actr_inputs()]
    with torch.no_grad():
        y = model(v_feat=v_feat, grid=grid, i_feats=[i_feat], lidar_grid=lidar_grid, v_i_feat=v_i_feat)
    names = np.array(sorted(shapes))
    save("actr.npz", out=y.numpy(), param_names=names,
         param_shapes=np.array([str(shapes[k]) for k in names]), n_params=np.array(sum(int(np.prod(s)) for s in shapes.values())))


# ------------------------------------------------------------------ CenterPoint fusion adapter (a7-a9)
FUS = dict(batch=2, pc_range=[-9.6, -9.6, -5.0, 9.6, 9.6, 3.0], voxel_size=[0.075, 0.075, 0.2],
           raw_hw=(240, 320), image_scale=2.0 / 3.0, img_hw=(160, 213), feat_hw=(40, 54), focal=250.0,
           depth_thres={'CAM_FRONT': 1, 'CAM_FRONT_LEFT': 0, 'CAM_FRONT_RIGHT': 0, 'CAM_BACK': 0.5,
                        'CAM_BACK_LEFT': 0, 'CAM_BACK_RIGHT': 0})
FUS_IFAT = dict(fusion_method='Basicgate_patch_iv_multivoxel', img_num_channel=256, pts_num_channel=128,
                voxel_feat_channel=[32, 64, 128], voxel_idx=[0, 2])
FUS_LT = dict(npoint=2048, radius=2.0, nsample=32, num_layers=2, attn_feat_agg_method='unique',
              feat_agg_method='replace')


def fusion_voxel_sets():
    """Active voxel coordinates of x_conv2 / x_conv3 / x_conv4 for two synthetic sweeps on the reduced
    grid (256 x 256 x 41 -> strides 2, 4, 8), through the oracle's rulebook chain."""
    from oracle import oracle as orc
    coors = []
    for b in range(FUS["batch"]):
        pts = synth.nusc_sweep(seed=40 + b)
        _, c, _ = orc.hard_voxelize(pts, FUS["voxel_size"], FUS["pc_range"], 10, 120000)
        coors.append(np.concatenate([np.full((len(c), 1), b, np.int32), c], 1))
    ind = np.concatenate(coors)
    shape = [41, 256, 256]
    out = []
    for pad in ([1, 1, 1], [1, 1, 1], [0, 1, 1]):
        o, _, _, shape = orc.get_indice_pairs(ind, FUS["batch"], shape, [3, 3, 3], [2, 2, 2], pad, [1, 1, 1], 0)
        order = np.lexsort((o[:, 3], o[:, 2], o[:, 1], o[:, 0]))   # (b,z,y,x)-sorted, as spconv's GPU path emits
        ind = np.ascontiguousarray(o[order])
        out.append(ind)
    return out


def fusion_inputs():
    sets = fusion_voxel_sets()
    feats = [detgen.randn("fus_feat%d" % i, (len(s), c)) for i, (s, c) in enumerate(zip(sets, [32, 64, 128]))]
    cams = synth.nusc_cameras(image_hw=FUS["raw_hw"], focal=FUS["focal"])
    B = FUS["batch"]
    img = {name.lower(): detgen.randn("fus_img_" + name, (B, 256) + FUS["feat_hw"]) for name in synth.NUSC_CAMS}
    return sets, feats, cams, img


def fusion_aug_inv():
    """aug_matrix_inv records of the two samples in the pipeline's format (preprocess.py:309-354): flip, rotate, rescale
    as TRANSPOSED inverse matrices for row vectors, translate as the negated noise."""
    out = []
    for b in range(FUS["batch"]):
        a = -(0.17 - 0.3 * b)
        c, s_ = np.cos(a), np.sin(a)
        sc = 1.0 / (1.05 - 0.08 * b)
        out.append(dict(flip=np.array([[[1, -1][b == 0], 0, 0], [0, [1, -1][b == 1], 0], [0, 0, 1]], np.float32),
                        rotate=np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], np.float32),
                        rescale=np.array([[sc, 0, 0], [0, sc, 0], [0, 0, sc]], np.float32),
                        translate=-np.array([0.21, -0.13, 0.05], np.float32) * (b + 1)))
    return out


def gen_fusion():
    """Reference VoxelWithPointProjection.forward (fuse_mode 'pfat', ACTR + ifat gate) on CPU.
    Environment shims only: `.cuda()` -> identity, torch.tensor(device='cuda') -> cpu, and kornia
    (absent, unpinned by the reference) restated from its published implementation
    (kornia.geometry.linalg.transform_points / conversions.convert_points_{to,from}_homogeneous, 0.6.x)."""
    import torch.nn.functional as F

    def to_h(p):
        return F.pad(p, [0, 1], "constant", 1.0)

    def from_h(p, eps=1e-8):
        z = p[..., -1:]
        scale = torch.where(z.abs() > eps, 1.0 / (z + eps), torch.ones_like(z))
        return scale * p[..., :-1]

    def transform_points(trans_01, points_1):
        shp = list(points_1.shape)
        p = points_1.reshape(-1, shp[-2], shp[-1])
        t = trans_01.reshape(-1, trans_01.shape[-2], trans_01.shape[-1])
        t = torch.repeat_interleave(t, repeats=p.shape[0] // t.shape[0], dim=0)
        o = from_h(torch.bmm(to_h(p), t.permute(0, 2, 1)))
        shp[-2], shp[-1] = o.shape[-2], o.shape[-1]
        return o.reshape(shp)

    _stub("kornia")
    _stub("kornia.utils")
    _stub("kornia.utils.grid", create_meshgrid3d=None)
    _stub("kornia.geometry")
    _stub("kornia.geometry.linalg", transform_points=transform_points)
    _stub("kornia.geometry.conversions", convert_points_to_homogeneous=to_h, convert_points_from_homogeneous=from_h)
    R = "/root/reference/CenterPoint/det3d"
    for pkg, path in [("det3d.models.fusion", R + "/models/fusion"), ("det3d.models.utils", R + "/models/utils"),
                      ("det3d.models.losses", R + "/models/losses"), ("det3d.core", R + "/core"),
                      ("det3d.datasets", R + "/datasets"), ("det3d.datasets.nuscenes", R + "/datasets/nuscenes")]:
        _stub(pkg).__path__ = [path]

    class _Reg:
        def register_module(self, cls):
            return cls

    _stub("det3d.models.registry", FUSION=_Reg())
    _stub("det3d.models.losses.auxseg_loss", SEGLOSS=None)
    _stub("det3d.core.bbox", box_np_ops=None)
    _stub("det3d.core.bbox.box_np_ops")
    _stub("det3d.datasets.nuscenes.nusc_common", get_lidar2cam_matrix=None, view_points=None)
    # pts2img (attention.py:422-468) writes duplicate pixels with index_put_: the winner is order
    # dependent (racy on the reference's GPU path and on a multi-threaded CPU).  The fixture is taken
    # with ONE CPU thread = sequential execution = last writer wins.
    torch.set_num_threads(1)
    torch.Tensor.cuda = lambda self, *a, **k: self
    _orig_tensor = torch.tensor
    torch.tensor = lambda *a, **k: _orig_tensor(*a, **{kk: ("cpu" if kk == "device" and str(v).startswith("cuda") else v)
                                                         for kk, v in k.items()})
    try:
        vwp = importlib.import_module("det3d.models.fusion.voxel_with_point_projection")
        sets, feats, cams, img = fusion_inputs()
        mod = vwp.VoxelWithPointProjection(
            fuse_mode='pfat', interpolate=False, voxel_size=FUS["voxel_size"], pc_range=FUS["pc_range"],
            image_list=synth.NUSC_CAMS, image_scale=FUS["image_scale"], depth_thres=FUS["depth_thres"],
            pfat_cfg=Cfg(ACTR_CFG), lt_cfg=Cfg(FUS_LT), ifat_cfg=Cfg(FUS_IFAT), model_name='ACTR').eval()
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        sd = detgen.det_state_dict(shapes)
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        B = FUS["batch"]

        class SpT:
            def __init__(self, f, i):
                self.features, self.indices = torch.from_numpy(f.copy()), torch.from_numpy(i.copy())

        xs = [SpT(f, i) for f, i in zip(feats, sets)]
        H, W = FUS["img_hw"]
        batch_dict = {
            'image_shape': {n.lower(): torch.tensor([[H, W, 3]] * B) for n in synth.NUSC_CAMS},
            'img_feat': {'layer1_ori_feat2d': {k: torch.from_numpy(v) for k, v in img.items()}},
            'calib': {},
        }
        for n in synth.NUSC_CAMS:
            key = n.lower().lstrip('cam_')
            T, K = cams[n]
            batch_dict['calib']['lidar2cam_' + key] = torch.from_numpy(np.stack([T] * B))
            batch_dict['calib']['cam_intrinsic_' + key] = torch.from_numpy(np.stack([K] * B))
        with torch.no_grad():
            out = mod(batch_dict, {}, encoded_voxel_list=xs, layer_name='layer1_ori', fuse_mode='pfat',
                      d_factor_list=[2, 4, 8])
        # per (b, cam) visible-voxel counts for diagnosis: recompute through the reference projector
        counts = np.zeros((B, 6), np.int64)
        for ci, n in enumerate(synth.NUSC_CAMS):
            pd = mod.point_projector(voxel_coords=torch.from_numpy(sets[2]).float(), image_scale=FUS["image_scale"],
                                     batch_dict=batch_dict, cam_key=n.lower(), d_factor=8)
            counts[:, ci] = pd['point_mask'].sum(1).numpy()
        # second case: 3-D augmentation records to undo before the projection (point_to_image_projection.py:121-128)
        batch_dict['aug_matrix_inv'] = fusion_aug_inv()
        xs = [SpT(f, i) for f, i in zip(feats, sets)]
        with torch.no_grad():
            out_aug = mod(batch_dict, {}, encoded_voxel_list=xs, layer_name='layer1_ori', fuse_mode='pfat',
                          d_factor_list=[2, 4, 8])
        counts_aug = np.zeros((B, 6), np.int64)
        for ci, n in enumerate(synth.NUSC_CAMS):
            pd = mod.point_projector(voxel_coords=torch.from_numpy(sets[2]).float(), image_scale=FUS["image_scale"],
                                     batch_dict=batch_dict, cam_key=n.lower(), d_factor=8)
            counts_aug[:, ci] = pd['point_mask'].sum(1).numpy()
        save("fusion_cp_aug.npz", out=out_aug.features.numpy(), counts=counts_aug)
        names = np.array(sorted(shapes))
        save("fusion_cp.npz", coords2=sets[0].astype(np.int16), coords3=sets[1].astype(np.int16),
             coords4=sets[2].astype(np.int16), out=out.features.numpy(), counts=counts, param_names=names,
             param_shapes=np.array([str(shapes[k]) for k in names]))
    finally:
        torch.tensor = _orig_tensor


# ------------------------------------------------------------------ LocalTransformer (a13)
LT_DIMS = dict(B=2, N=300, C=32, npoint=64, radius=2.0, nsample=8, num_layers=2)


def lt_inputs():
    d = LT_DIMS
    xyz = detgen.rand("lt_xyz", (d["B"], d["N"], 3), -6, 6)
    feat = detgen.randn("lt_feat", (d["B"], d["C"], d["N"]))
    xyz[1, 250:] = 0          # zero-padded tail, as the per-camera query batches have
    feat[1, :, 250:] = 0
    return xyz, feat


def gen_local_transformer():
    """Reference LocalTransformer.forward (pointformer.py:349-380) on CPU.  Its four CUDA-only index
    ops have no CPU implementation in the reference; they are bound to the oracle, which gen_pointops()
    pins to the reference's own unit-test literals.  mmcv.cnn.ConvModule (absent) is restated for the
    two configurations used (conv [+BN2d] [+ReLU]; attribute names conv/bn/activate)."""
    from oracle import oracle as orc
    import_reference_actr()

    class ConvModule(torch.nn.Module):
        def __init__(self, cin, cout, k, norm_cfg=None, act_cfg=dict(type="ReLU")):
            super().__init__()
            self.conv = torch.nn.Conv2d(cin, cout, k, bias=norm_cfg is None)
            self.bn = torch.nn.BatchNorm2d(cout) if norm_cfg is not None else None
            self.activate = torch.nn.ReLU(inplace=True) if act_cfg is not None else None
            if self.bn is None:
                del self.bn
            if self.activate is None:
                del self.activate

        def forward(self, x):
            x = self.conv(x)
            if hasattr(self, "bn"):
                x = self.bn(x)
            if hasattr(self, "activate"):
                x = self.activate(x)
            return x

    pf = importlib.import_module("det3d.models.model_utils.pointformer")
    pf.ConvModule = ConvModule
    pf.gather_points = lambda f, i: torch.from_numpy(orc.gather_points(f.numpy(), i.numpy()))

    class Sampler(torch.nn.Module):
        def __init__(self, num_point, mods):
            super().__init__()
            self.m = num_point[0]

        def forward(self, xyz, feats):
            return torch.from_numpy(orc.furthest_point_sample(xyz.numpy(), self.m))

    class Grouper(torch.nn.Module):
        def __init__(self, radius, nsample, **kw):
            super().__init__()
            self.r, self.ns = radius, nsample

        def forward(self, xyz, new_xyz, feats):
            idx = orc.ball_query(0.0, self.r, self.ns, xyz.numpy(), new_xyz.numpy())
            gx = orc.group_points(xyz.transpose(1, 2).contiguous().numpy(), idx)
            gf = orc.group_points(feats.numpy(), idx)
            return torch.from_numpy(gf), torch.from_numpy(gx), torch.from_numpy(idx)

    pf.Points_Sampler, pf.QueryAndGroup = Sampler, Grouper
    # torch >= 2 passes is_causal to encoder layers; the reference layer (torch 1.x era) does not take it
    _fw = pf.TransformerEncoderLayerPreNorm.forward
    pf.TransformerEncoderLayerPreNorm.forward = lambda self, src, src_mask=None, src_key_padding_mask=None, **kw: \
        _fw(self, src, src_mask, src_key_padding_mask)
    d = LT_DIMS
    torch.set_num_threads(1)
    m = pf.LocalTransformer(d["npoint"], d["radius"], d["nsample"], d["C"], d["C"], num_layers=d["num_layers"],
                            attn_feat_agg_method="unique", feat_agg_method="replace").eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = detgen.det_state_dict(shapes)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    xyz, feat = lt_inputs()
    with torch.no_grad():
        y = m(torch.from_numpy(xyz), torch.from_numpy(feat.copy()))
    names = np.array(sorted(shapes))
    save("local_transformer.npz", out=y.numpy(), param_names=names,
         param_shapes=np.array([str(shapes[k]) for k in names]))


# ------------------------------------------------------------------ point ops: the reference's own test vectors
def gen_pointops():
    """Run the reference's GPU-only unit tests for FPS / ball query / grouping / gathering
    (TF/tests/test_models/test_common_modules/test_pointnet_ops.py:9-74,126-238) with `mmdet3d.ops`
    bound to the ORACLE (so a wrong oracle fails right here) and record every (inputs, expected)
    pair the tests hold as literals.  The recorded tensors are data; no test source is stored."""
    from oracle import oracle as orc
    rec = []

    def wrap(name, fn):
        def f(*a):
            out = fn(*a)
            rec.append([name, [np.asarray(x.numpy() if torch.is_tensor(x) else x) for x in a], None])
            return out
        return f

    def fps(xyz, m):
        return torch.from_numpy(orc.furthest_point_sample(xyz.numpy(), m))

    def bq(min_r, max_r, ns, xyz, new_xyz):
        return torch.from_numpy(orc.ball_query(min_r, max_r, ns, xyz.numpy(), new_xyz.numpy()))

    def grp(feat, idx):
        return torch.from_numpy(orc.group_points(feat.numpy(), idx.numpy()))

    def gat(feat, idx):
        return torch.from_numpy(orc.gather_points(feat.numpy(), idx.numpy()))

    ops_mod = _stub("mmdet3d.ops", ball_query=wrap("ball_query", bq), furthest_point_sample=wrap("fps", fps),
                    furthest_point_sample_with_dist=None, gather_points=wrap("gather_points", gat),
                    grouping_operation=wrap("group_points", grp), knn=None, three_interpolate=None, three_nn=None)
    _stub("mmdet3d").ops = ops_mod
    src = open("/root/reference/TransFusion/tests/test_models/test_common_modules/test_pointnet_ops.py").read()
    ns = {}
    _cuda, _avail, _all, _allclose = torch.Tensor.cuda, torch.cuda.is_available, torch.all, torch.allclose
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.is_available = lambda: True

    def rec_all(x):
        return _all(x)

    try:
        exec(compile(src, "test_pointnet_ops", "exec"), ns)
        out = {}
        for tname in ("test_fps", "test_ball_query", "test_grouping_points", "test_gather_points"):
            n0 = len(rec)
            # capture `expected_*` locals by tracing the assert comparisons
            import sys as _sys
            captured = []

            def tracer(frame, event, arg):
                if event == "return" and frame.f_code.co_name == tname:
                    captured.append({k: v for k, v in frame.f_locals.items() if torch.is_tensor(v)})
                return tracer
            _sys.settrace(tracer)
            try:
                ns[tname]()          # asserts inside: the oracle must reproduce the reference's literals
            finally:
                _sys.settrace(None)
            loc = captured[-1]
            for k, v in loc.items():
                out["%s__%s" % (tname, k)] = v.numpy()
        save("pointops_tests.npz", **out)
    finally:
        torch.Tensor.cuda, torch.cuda.is_available = _cuda, _avail


HEAD_TASKS = [dict(num_class=1, class_names=["car"]), dict(num_class=2, class_names=["truck", "construction_vehicle"]),
              dict(num_class=2, class_names=["bus", "trailer"]), dict(num_class=1, class_names=["barrier"]),
              dict(num_class=2, class_names=["motorcycle", "bicycle"]),
              dict(num_class=2, class_names=["pedestrian", "traffic_cone"])]
HEAD_COMMON = {'reg': (2, 2), 'height': (1, 2), 'dim': (3, 2), 'rot': (2, 2), 'vel': (2, 2)}
HEAD_SHAPE = (2, 512, 12, 14)
# nuScenes test_cfg of the 3D-DF config (nusc_centerpoint_voxelnet_0075voxel_fix_bn_z_multimodal_pfat_hybrid7_ifat.py:
# 145-159) with a smaller pre_max (so that the 168-pixel test map exercises the truncation) and a z range that masks
HEAD_TEST_CFG = dict(post_center_limit_range=[-61.2, -61.2, -0.6, 61.2, 61.2, 0.7], max_per_img=500,
                     nms=dict(use_rotate_nms=True, use_multi_class_nms=False, nms_pre_max_size=60,
                              nms_post_max_size=83, nms_iou_threshold=0.2),
                     score_threshold=0.1, pc_range=[-54, -54], out_size_factor=8, voxel_size=[0.075, 0.075])


def head_bias_shift(sd):
    """Liven the detgen weights up: larger final convolutions (spread-out scores, heights, sizes) and lower heat-map
    logits, so that the score threshold, the z range and the NMS all have something to decide."""
    for k in sd:
        if k.endswith(".3.weight"):
            sd[k] = sd[k] * (12.0 if ".hm." in k else 6.0)
        if ".hm.3.bias" in k:
            sd[k] = sd[k] - 1.5
    return sd


def import_reference_centerhead():
    """CP/det3d/models/bbox_heads/center_head.py with import stubs (registry, Sequential and box_torch_ops are the
    reference's own files; numba.jit -> identity; kaiming_init -> no-op, the weights are overwritten)."""
    import importlib.util
    R = "/root/reference/CenterPoint/det3d"

    def load_file(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[name] = m
        spec.loader.exec_module(m)
        return m
    for pkg, path in [("det3d", R), ("det3d.models", R + "/models"), ("det3d.models.bbox_heads", R + "/models/bbox_heads"),
                      ("det3d.models.losses", R + "/models/losses"), ("det3d.models.utils", R + "/models/utils"),
                      ("det3d.core", R + "/core"), ("det3d.core.utils", R + "/core/utils"),
                      ("det3d.core.bbox", R + "/core/bbox"), ("det3d.utils", R + "/utils"),
                      ("det3d.torchie", R + "/torchie"), ("det3d.ops", R + "/ops")]:
        _stub(pkg).__path__ = [path]
    _stub("numba", jit=lambda *a, **k: (lambda f: f))
    reg = load_file("det3d.utils.registry", R + "/utils/registry.py")
    sys.modules["det3d.utils"].Registry = reg.Registry
    sys.modules["det3d.utils"].build_from_cfg = reg.build_from_cfg
    _stub("det3d.torchie.cnn", kaiming_init=lambda m, **k: None)
    misc = load_file("det3d.models.utils.misc", R + "/models/utils/misc.py")
    sys.modules["det3d.models.utils"].Sequential = misc.Sequential
    bto = load_file("det3d.core.bbox.box_torch_ops", R + "/core/bbox/box_torch_ops.py")
    sys.modules["det3d.core"].box_torch_ops = bto
    ch = importlib.import_module("det3d.models.bbox_heads.center_head")

    class _Nms:                                   # the CUDA extension's entry point on the reference's CPU IoU
        @staticmethod
        def nms_gpu(boxes, keep, thresh):
            b = boxes.numpy()
            iou = ref.boxes_iou_bev_cpu(b, b)
            removed = np.zeros(len(b), bool)
            n = 0
            for i in range(len(b)):               # iou3d_nms.cpp:118-133
                if removed[i]:
                    continue
                keep[n] = i
                n += 1
                removed[i + 1:] |= iou[i, i + 1:] > thresh
            return n
    bto.iou3d_nms_cuda = _Nms
    return ch, bto


def gen_centerhead():
    """Reference CenterHead forward + predict (center_head.py:237-247,302-501) on a small BEV map."""
    from oracle import oracle as orc
    ch, _ = import_reference_centerhead()
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        head = ch.CenterHead(in_channels=512, tasks=HEAD_TASKS, dataset='nuscenes', weight=0.25,
                             code_weights=[1.0] * 10, common_heads=dict(HEAD_COMMON), share_conv_channel=64,
                             dcn_head=False)
    shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    sd = head_bias_shift(detgen.det_state_dict(shapes))
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    head.eval()
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for k in range(400):
            x = torch.from_numpy(detgen.randn("head_x_%d" % k, HEAD_SHAPE))
            with torch.no_grad():
                preds = head(x)
            stats = {}
            pn = [{kk: vv.numpy().copy() for kk, vv in p.items()} for p in preds]
            orc.centerhead_predict(pn, HEAD_TEST_CFG, [1, 2, 2, 1, 2, 2], margin_out=stats)
            if stats["score_gap"] > 2e-6 and stats["thr_gap"] > 1e-5 and stats["range_gap"] > 1e-4 and stats["iou_close"] == 0:
                break
        else:
            raise RuntimeError("no tie-free head input found")
        with torch.no_grad():
            dets = head.predict({}, [{kk: vv.clone() for kk, vv in p.items()} for p in preds],
                                Cfg(HEAD_TEST_CFG, nms=Cfg(HEAD_TEST_CFG["nms"])))
    finally:
        torch.Tensor.cuda = _cuda
    out = dict(seed=np.int64(k), keys=np.array(sorted(shapes)), margins=np.array([stats["score_gap"], stats["thr_gap"],
                                                                                 stats["range_gap"]]))
    for t, p in enumerate(pn):
        for name, v in p.items():
            out["chk_%d_%s" % (t, name)] = np.array([v.sum(dtype=np.float64), np.abs(v).sum(dtype=np.float64)])
    for i, d in enumerate(dets):
        out["boxes_%d" % i] = d["box3d_lidar"].numpy()
        out["scores_%d" % i] = d["scores"].numpy()
        out["labels_%d" % i] = d["label_preds"].numpy()
    print("seed", k, stats, [len(d["scores"]) for d in dets])
    save("centerhead.npz", **out)


TFH_SHAPE = (2, 512, 20, 22)                       # [B, 2 x 256 neck channels, H, W]
TFH_KW = dict(num_proposals=24, auxiliary=True, in_channels=512, hidden_channel=128, num_classes=10,
              num_decoder_layers=1, num_heads=8, learnable_query_pos=False, initialize_by_heatmap=True,
              nms_kernel_size=3, ffn_channel=256, dropout=0.1, bn_momentum=0.1, activation='relu',
              common_heads=dict(center=(2, 2), height=(1, 2), dim=(3, 2), rot=(2, 2), vel=(2, 2)))
# transfusion_nusc_voxel_F.py:244-300 on a small map: grid_size / out_size_factor give the 20 x 22 BEV grid
# (grid_size[0] sizes the first mesh axis = rows), and the centre range is narrowed so that the range mask decides.
TFH_CODER = dict(pc_range=[-6.0, -6.6], voxel_size=[0.075, 0.075], out_size_factor=8,
                 post_center_range=[-5.2, -5.9, -10.0, 5.0, 5.6, 10.0], score_threshold=0.0, code_size=10)
TFH_TEST_CFG = dict(dataset='nuScenes', grid_size=[160, 176, 40], out_size_factor=8, pc_range=[-6.0, -6.6],
                    voxel_size=[0.075, 0.075], nms_type=None)


def tfh_weight_shift(sd):
    """detgen weights with a wider heat map (final 3x3 conv) and wider box regressions."""
    for k in sd:
        if k == "heatmap_head.1.weight":
            sd[k] = sd[k] * 10.0
        if k.startswith("prediction_heads.") and k.endswith(".1.weight"):
            sd[k] = sd[k] * 4.0
    return sd


def import_reference_transfusion_head():
    """TF/mmdet3d/models/dense_heads/transfusion_head.py and core/bbox/coders/transfusion_bbox_coder.py (the
    reference's own files) under import stubs for the absent mmcv / mmdet packages: ConvModule is restated for the
    two configurations the head uses (Conv1d/2d [+ BN] + ReLU, attribute names conv / bn / activate, bias = no norm),
    build_conv_layer maps the cfg type onto torch.nn, multi_apply is mmdet's map-and-transpose."""
    import importlib.util
    R = "/root/reference/TransFusion/mmdet3d"

    class ConvModule(torch.nn.Module):
        def __init__(self, cin, cout, kernel_size, stride=1, padding=0, bias='auto', conv_cfg=None, norm_cfg=None):
            super().__init__()
            conv = getattr(torch.nn, (conv_cfg or dict(type='Conv2d'))['type'])
            self.conv = conv(cin, cout, kernel_size, stride=stride, padding=padding,
                             bias=(norm_cfg is None) if bias == 'auto' else bias)
            if norm_cfg is not None:
                self.bn = {'BN1d': torch.nn.BatchNorm1d, 'BN2d': torch.nn.BatchNorm2d}[norm_cfg['type']](cout)
            self.activate = torch.nn.ReLU(inplace=True)

        def forward(self, x):
            x = self.conv(x)
            if hasattr(self, "bn"):
                x = self.bn(x)
            return self.activate(x)

    def build_conv_layer(cfg, *a, **k):
        return getattr(torch.nn, cfg['type'])(*a, **k)

    class Reg:
        def __init__(self):
            self.d = {}

        def register_module(self, *a, **k):
            def deco(c):
                self.d[c.__name__] = c
                return c
            return deco
    heads, coders = Reg(), Reg()
    _stub("mmcv")
    _stub("mmcv.cnn", ConvModule=ConvModule, build_conv_layer=build_conv_layer, kaiming_init=lambda m, **k: None)
    _stub("mmcv.runner", force_fp32=lambda *a, **k: (lambda f: f))
    _stub("mmdet")
    _stub("mmdet.core.bbox", BaseBBoxCoder=object)
    _stub("mmdet.core.bbox.builder", BBOX_CODERS=coders)
    spec = importlib.util.spec_from_file_location("tf_coder", R + "/core/bbox/coders/transfusion_bbox_coder.py")
    cm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cm)

    def multi_apply(func, *args):
        return tuple(map(list, zip(*map(func, *args))))

    def build_bbox_coder(cfg):
        cfg = dict(cfg)
        return coders.d[cfg.pop('type')](**cfg)
    _stub("mmdet.core", build_bbox_coder=build_bbox_coder, multi_apply=multi_apply, build_assigner=None,
          build_sampler=None, AssignResult=None)
    _stub("mmdet3d")
    _stub("mmdet3d.core", circle_nms=None, draw_heatmap_gaussian=None, gaussian_radius=None, xywhr2xyxyr=None,
          limit_period=None, PseudoSampler=None, Box3DMode=None, LiDARInstance3DBoxes=None)
    _stub("mmdet3d.core.bbox")
    _stub("mmdet3d.core.bbox.structures", rotation_3d_in_axis=None)
    mb = _stub("mmdet3d.models.builder", HEADS=heads, build_loss=lambda cfg: None)
    _stub("mmdet3d.models", builder=mb)
    _stub("mmdet3d.models.utils", clip_sigmoid=None)
    _stub("mmdet3d.models.fusion_layers", apply_3d_transformation=None)
    _stub("mmdet3d.ops")
    _stub("mmdet3d.ops.iou3d")
    _stub("mmdet3d.ops.iou3d.iou3d_utils", nms_gpu=None)
    _stub("mmdet3d.ops.roiaware_pool3d", points_in_boxes_batch=None)
    spec = importlib.util.spec_from_file_location("tf_head", R + "/models/dense_heads/transfusion_head.py")
    hm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hm)
    return hm


def gen_transfusion_head():
    """Reference TransFusionHead.forward + get_bboxes (transfusion_head.py:797-1045,1285-1376), LiDAR-only branch
    (fuse_img=False, as in transfusion_nusc_voxel_F.py:244-300), eval mode, CPU.  The input seed is searched until the
    proposal selection survives a 1e-5 relative perturbation of the input (the top-k / local-maximum decisions have
    margins), so that an fp32 implementation with a different summation order picks the same proposals."""
    hm = import_reference_transfusion_head()
    head = hm.TransFusionHead(loss_cls=dict(use_sigmoid=True), loss_iou=dict(), loss_bbox=dict(), loss_heatmap=dict(),
                              train_cfg=None, test_cfg=dict(TFH_TEST_CFG),
                              bbox_coder=dict(type='TransFusionBBoxCoder', **TFH_CODER), **TFH_KW)
    shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    sd = tfh_weight_shift(detgen.det_state_dict(shapes))
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    head.eval()
    for k in range(200):
        x = torch.from_numpy(detgen.randn("tfh_x_%d" % k, TFH_SHAPE))
        with torch.no_grad():
            res = head([x], None, [{}])
            labels = head.query_labels.clone()
            picks = []
            for j in range(3):
                noise = torch.from_numpy(detgen.randn("tfh_n_%d_%d" % (k, j), TFH_SHAPE))
                r2 = head([x * (1 + 1e-5 * noise)], None, [{}])
                picks.append((r2[0][0]['center'] - res[0][0]['center']).abs().max().item() < 1e-2 and
                             torch.equal(head.query_labels, labels))
        if all(picks):
            break
    else:
        raise RuntimeError("no stable TransFusionHead input found")
    with torch.no_grad():
        res = head([x], None, [{}])
        labels = head.query_labels.clone()
        # get_bboxes asserts batch size 1 at its end (transfusion_head.py:1366-1367); the per-sample part is taken
        # sample by sample exactly as a batch-1 call would see it
        dets = []
        for b in range(TFH_SHAPE[0]):
            head.query_labels = labels[b:b + 1]
            one = ([{kk: vv[b:b + 1].clone() for kk, vv in res[0][0].items()}],)
            box, score, lab = head.get_bboxes(one, [dict(box_type_3d=lambda t, box_dim: t)])[0]
            dets.append((box.numpy(), score.numpy(), lab.numpy()))
    out = dict(seed=np.int64(k), keys=np.array(sorted(shapes)), query_labels=labels.numpy())
    for name, v in res[0][0].items():
        out["pred_" + name] = v.numpy()
    for b, (box, score, lab) in enumerate(dets):
        out["boxes_%d" % b], out["scores_%d" % b], out["labels_%d" % b] = box, score, lab
    print("seed", k, {n: tuple(v.shape) for n, v in res[0][0].items()}, [len(d[1]) for d in dets])
    save("transfusion_head.npz", **out)


# ------------------------------------------------------------------ TransFusionHead.loss (round 3)
TFL_SHAPE = (2, 512, 20, 20)                       # square map: the reference's target heat map is [C, y_len, x_len]
TFL_RANGE = [-6.0, -6.0, -5.0, 6.0, 6.0, 3.0]      # while forward_single lays the BEV grid out as [x_len, y_len]
TFL_CODER = dict(pc_range=TFL_RANGE[:2], voxel_size=[0.075, 0.075], out_size_factor=8,
                 post_center_range=[-5.2, -5.9, -10.0, 5.0, 5.6, 10.0], score_threshold=0.0, code_size=10)
TFL_TEST_CFG = dict(dataset='nuScenes', grid_size=[160, 160, 40], out_size_factor=8, pc_range=TFL_RANGE[:2],
                    voxel_size=[0.075, 0.075], nms_type=None)
# transfusion_nusc_voxel_L.py:217-234
TFL_TRAIN_CFG = dict(dataset='nuScenes',
                     assigner=dict(type='HungarianAssigner3D', iou_calculator=dict(type='BboxOverlaps3D', coordinate='lidar'),
                                   cls_cost=dict(type='FocalLossCost', gamma=2, alpha=0.25, weight=0.15),
                                   reg_cost=dict(type='BBoxBEVL1Cost', weight=0.25), iou_cost=dict(type='IoU3DCost', weight=0.25)),
                     pos_weight=-1, gaussian_overlap=0.1, min_radius=2, grid_size=[160, 160, 40], voxel_size=[0.075, 0.075, 0.2],
                     out_size_factor=8, code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2],
                     point_cloud_range=TFL_RANGE)
TFL_LOSSES = dict(loss_cls=dict(type='FocalLoss', use_sigmoid=True, gamma=2, alpha=0.25, reduction='mean', loss_weight=1.0),
                  loss_bbox=dict(type='L1Loss', reduction='mean', loss_weight=0.25),
                  loss_heatmap=dict(type='GaussianFocalLoss', reduction='mean', loss_weight=1.0))


class CfgDict(dict):
    """mmcv.ConfigDict as the head uses it: attribute access, nested."""

    def __init__(self, d):
        super().__init__({k: CfgDict(v) if isinstance(v, dict) else v for k, v in d.items()})

    __getattr__ = dict.__getitem__


def import_reference_transfusion_loss():
    """The reference's own transfusion_head.py, hungarian_assigner.py, iou3d_calculator.py, core/bbox/structures/*,
    core/utils/gaussian.py, models/utils/clip_sigmoid.py and transfusion_bbox_coder.py, imported from /root/reference
    under synthetic parent packages.  What the reference takes from the absent mmdet 2.10.0 / mmcv is restated in
    tests/golden/mmdet_restated.py (losses, FocalLossCost, AssignResult, PseudoSampler) and above (ConvModule);
    the CUDA-only iou3d_cuda.boxes_overlap_bev_gpu is bound to the oracle's restatement of that kernel
    (oracle/df3d_oracle.c orc_tf_box_overlap; pinned against the reference's kernel on the GPU box)."""
    import importlib.util
    import mmdet_restated as mr
    from oracle import oracle as orc
    hm0 = import_reference_transfusion_head()             # ConvModule / registry stubs of the forward golden
    R = "/root/reference/TransFusion/mmdet3d"
    heads = sys.modules["mmdet3d.models.builder"].HEADS

    def overlap_bev(a, b, out):
        out.copy_(torch.from_numpy(orc.tf_boxes_overlap_bev(a.detach().numpy(), b.detach().numpy())))
    for name, path in [("mmdet3d", R), ("mmdet3d.core", R + "/core"), ("mmdet3d.core.bbox", R + "/core/bbox"),
                       ("mmdet3d.core.utils", R + "/core/utils"), ("mmdet3d.core.bbox.assigners", R + "/core/bbox/assigners"),
                       ("mmdet3d.core.bbox.iou_calculators", R + "/core/bbox/iou_calculators"), ("mmdet3d.ops", R + "/ops"),
                       ("mmdet3d.models.utils", R + "/models/utils")]:
        _stub(name).__path__ = [path]
    _stub("mmdet3d.ops.iou3d", iou3d_cuda=types.SimpleNamespace(boxes_overlap_bev_gpu=overlap_bev))
    _stub("mmdet3d.ops.roiaware_pool3d", points_in_boxes_gpu=None, points_in_boxes_batch=None)
    sys.modules["mmdet3d.ops"].points_in_boxes_batch = None
    sys.modules.pop("mmdet3d.core.bbox.structures", None)   # the forward golden's placeholder
    importlib.import_module("mmdet3d.core.points")
    st = importlib.import_module("mmdet3d.core.bbox.structures")
    gs = importlib.import_module("mmdet3d.core.utils.gaussian")
    cs = importlib.import_module("mmdet3d.models.utils.clip_sigmoid")

    class Reg:
        def __init__(self):
            self.d = {}

        def register_module(self, *a, **k):
            def deco(c):
                self.d[c.__name__] = c
                return c
            return deco

        def build(self, cfg):
            cfg = dict(cfg)
            return self.d[cfg.pop('type')](**cfg)
    assigners, costs, ious, coders = Reg(), Reg(), Reg(), sys.modules["mmdet.core.bbox.builder"].BBOX_CODERS
    costs.d["FocalLossCost"] = mr.FocalLossCost
    _stub("mmdet.core.bbox", BaseBBoxCoder=object, bbox_overlaps=None)
    _stub("mmdet.core.bbox.builder", BBOX_ASSIGNERS=assigners, BBOX_CODERS=coders)
    _stub("mmdet.core.bbox.assigners", AssignResult=mr.AssignResult, BaseAssigner=object)
    _stub("mmdet.core.bbox.match_costs", build_match_cost=costs.build)
    _stub("mmdet.core.bbox.match_costs.builder", MATCH_COST=costs)
    _stub("mmdet.core.bbox.iou_calculators", build_iou_calculator=ious.build)
    _stub("mmdet.core.bbox.iou_calculators.builder", IOU_CALCULATORS=ious)
    importlib.import_module("mmdet3d.core.bbox.iou_calculators.iou3d_calculator")
    importlib.import_module("mmdet3d.core.bbox.assigners.hungarian_assigner")

    def multi_apply(func, *args):
        return tuple(map(list, zip(*map(func, *args))))

    def build_bbox_coder(cfg):
        cfg = dict(cfg)
        return coders.d[cfg.pop('type')](**cfg)
    _stub("mmdet.core", build_bbox_coder=build_bbox_coder, multi_apply=multi_apply, build_assigner=assigners.build,
          build_sampler=None, AssignResult=mr.AssignResult)
    core = sys.modules["mmdet3d.core"]
    core.__dict__.update(circle_nms=None, draw_heatmap_gaussian=gs.draw_heatmap_gaussian, gaussian_radius=gs.gaussian_radius,
                         xywhr2xyxyr=st.xywhr2xyxyr, limit_period=st.limit_period, PseudoSampler=mr.PseudoSampler,
                         Box3DMode=st.Box3DMode, LiDARInstance3DBoxes=st.LiDARInstance3DBoxes)
    mb = _stub("mmdet3d.models.builder", HEADS=heads, build_loss=mr.build_loss)
    _stub("mmdet3d.models", builder=mb)
    sys.modules["mmdet3d.models.utils"].clip_sigmoid = cs.clip_sigmoid
    _stub("mmdet3d.models.fusion_layers", apply_3d_transformation=None)
    _stub("mmdet3d.ops.iou3d.iou3d_utils", nms_gpu=None)
    spec = importlib.util.spec_from_file_location("tf_head_loss", R + "/models/dense_heads/transfusion_head.py")
    hm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hm)
    return hm, st


def tfl_gt_boxes(pred_boxes, b):
    """Ground truth of sample b: a few jittered copies of decoded proposals (non-zero 3-D IoU for the matcher) plus
    free boxes; (x, y, z_bottom, w, l, h, yaw, vx, vy) and a class per box."""
    rs = np.random.RandomState(700 + b)
    n_near, n_free = 4 + b, 3
    pick = rs.choice(len(pred_boxes), n_near, replace=False)
    near = pred_boxes[pick].copy()
    near[:, :2] += rs.normal(scale=0.25, size=(n_near, 2))
    near[:, 2] += rs.normal(scale=0.1, size=n_near)
    near[:, 3:6] *= np.exp(rs.normal(scale=0.15, size=(n_near, 3)))
    near[:, 6] += rs.normal(scale=0.2, size=n_near)
    free = np.zeros((n_free, 9))
    free[:, :2] = rs.uniform(-5, 5, size=(n_free, 2))
    free[:, 2] = rs.uniform(-2, 0, size=n_free)
    free[:, 3:6] = rs.uniform(0.5, 4.0, size=(n_free, 3))
    free[:, 6] = rs.uniform(-3.1, 3.1, size=n_free)
    free[:, 7:] = rs.normal(size=(n_free, 2))
    boxes = np.concatenate([near, free]).astype(np.float32)
    boxes[:, :2] = np.clip(boxes[:, :2], -5.6, 5.6)
    return boxes, rs.randint(0, 10, size=len(boxes)).astype(np.int64)


def gen_transfusion_head_loss():
    """Reference TransFusionHead.forward + get_targets + loss (transfusion_head.py:1048-1283) with the reference's
    HungarianAssigner3D (hungarian_assigner.py:100-160), BboxOverlaps3D, TransFusionBBoxCoder.encode, gaussian
    heat-map targets, and the gradient the summed losses send back; eval mode (no dropout, running BN statistics)."""
    hm, st = import_reference_transfusion_loss()
    kw = dict(TFH_KW)
    head = hm.TransFusionHead(train_cfg=CfgDict(TFL_TRAIN_CFG), test_cfg=dict(TFL_TEST_CFG), loss_iou=dict(type='VarifocalLoss'),
                              bbox_coder=dict(type='TransFusionBBoxCoder', **TFL_CODER), **TFL_LOSSES, **kw)
    shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    sd = tfh_weight_shift(detgen.det_state_dict(shapes))
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    head.eval()
    _cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for k in range(50):
            x = torch.from_numpy(detgen.randn("tfl_x_%d" % k, TFL_SHAPE)).requires_grad_(True)
            res = head([x], None, [{}])
            p = res[0][0]
            with torch.no_grad():
                dec = head.bbox_coder.decode(p['heatmap'].clone(), p['rot'].clone(), p['dim'].clone(), p['center'].clone(),
                                             p['height'].clone(), p['vel'].clone())
            gts = [tfl_gt_boxes(dec[b]['bboxes'].numpy(), b) for b in range(TFL_SHAPE[0])]
            gt_boxes = [st.LiDARInstance3DBoxes(torch.from_numpy(g[0]), box_dim=9) for g in gts]
            gt_labels = [torch.from_numpy(g[1]) for g in gts]
            # the matching must survive a small perturbation of the cost matrix (different summation orders / ulps)
            from scipy.optimize import linear_sum_assignment
            stable, costs = True, []
            for b in range(TFL_SHAPE[0]):
                a = head.bbox_assigner
                boxes = dec[b]['bboxes']
                cost = (a.cls_cost(p['heatmap'][b].detach().T, gt_labels[b]) + a.reg_cost(boxes, gt_boxes[b].tensor, head.train_cfg)
                        + a.iou_cost(a.iou_calculator(boxes, gt_boxes[b].tensor))).numpy()
                costs.append(cost)
                r0 = linear_sum_assignment(cost)
                for j in range(8):
                    noise = detgen.randn("tfl_noise_%d_%d_%d" % (k, b, j), cost.shape)
                    r1 = linear_sum_assignment(cost + 2e-4 * noise)
                    stable &= np.array_equal(r0[0], r1[0]) and np.array_equal(r0[1], r1[1])
            if stable:
                break
        else:
            raise RuntimeError("no stable matching found")
        targets = head.get_targets(gt_boxes, gt_labels, res[0])
        dense_logits = p['dense_heatmap'].detach().clone()       # loss() applies sigmoid_ in place to this prediction
        losses = head.loss(gt_boxes, gt_labels, res)
        total = sum(v for n, v in losses.items() if 'loss' in n)
        total.backward()
    finally:
        torch.Tensor.cuda = _cuda
    out = dict(seed=np.int64(k), keys=np.array(sorted(shapes)), query_labels=head.query_labels.numpy(),
               dense_logits=dense_logits.numpy(),
               labels=targets[0].numpy(), label_weights=targets[1].numpy(), bbox_targets=targets[2].numpy(),
               bbox_weights=targets[3].numpy(), ious=targets[4].numpy(), num_pos=np.int64(targets[5]),
               matched_ious=np.float64(targets[6]), heatmap=targets[7].numpy(),
               gx=np.array([x.grad.sum(dtype=torch.float64).item(), x.grad.abs().sum(dtype=torch.float64).item()]),
               gw=np.array([head.shared_conv.weight.grad.abs().sum(dtype=torch.float64).item(),
                            head.heatmap_head[1].bias.grad.abs().sum(dtype=torch.float64).item(),
                            head.prediction_heads[0].center[1].weight.grad.abs().sum(dtype=torch.float64).item(),
                            head.decoder[0].multihead_attn.in_proj_weight.grad.abs().sum(dtype=torch.float64).item()]),
               gx_slice=x.grad[:, :8].numpy())
    for b, (g, l) in enumerate(gts):
        out["gt_boxes_%d" % b], out["gt_labels_%d" % b], out["cost_%d" % b] = g, l, costs[b]
    for name, v in p.items():
        if name != 'dense_heatmap':
            out["pred_" + name] = v.detach().numpy()
    for name, v in losses.items():
        out["loss_" + name] = np.float64(v.item())
    print("seed", k, {n: float(v) for n, v in losses.items()}, "num_pos", targets[5], "matched_ious", targets[6])
    save("transfusion_head_loss.npz", **out)


def head_loss_example():
    """Assigner outputs for CenterHead.loss on the golden map (HEAD_SHAPE): per task a heat-map target with a few unit
    peaks, flat pixel indices, mask, category id and box codes of M = 6 object slots."""
    B, _, H, W = HEAD_SHAPE
    M, ex = 6, dict(hm=[], ind=[], mask=[], cat=[], anno_box=[])
    for t, task in enumerate(HEAD_TASKS):
        nc = task["num_class"]
        rs = np.random.RandomState(100 + t)
        hm = (rs.uniform(0, 1, size=(B, nc, H, W)) ** 6).astype(np.float32)
        ind = rs.randint(0, H * W, size=(B, M)).astype(np.int64)
        mask = (rs.uniform(size=(B, M)) < 0.7).astype(np.uint8)
        mask[0, 0] = 1
        cat = rs.randint(0, nc, size=(B, M)).astype(np.int64)
        for b in range(B):
            for m in range(M):
                if mask[b, m]:
                    hm[b, cat[b, m], ind[b, m] // W, ind[b, m] % W] = 1.0
        ex["hm"].append(hm), ex["ind"].append(ind), ex["mask"].append(mask), ex["cat"].append(cat)
        ex["anno_box"].append(rs.normal(size=(B, M, 10)).astype(np.float32))
    return ex


def gen_centerhead_loss():
    """Reference CenterHead.forward (train mode) + loss (center_head.py:237-298) and the gradient it sends back."""
    ch, _ = import_reference_centerhead()
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        head = ch.CenterHead(in_channels=512, tasks=HEAD_TASKS, dataset='nuscenes', weight=0.25,
                             code_weights=[1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.2, 0.2, 1.0, 1.0], common_heads=dict(HEAD_COMMON),
                             share_conv_channel=64, dcn_head=False)
    shapes = {k: tuple(v.shape) for k, v in head.state_dict().items()}
    sd = head_bias_shift(detgen.det_state_dict(shapes))
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    head.eval()                                               # running BN statistics: the forward is deterministic
    x = torch.from_numpy(detgen.randn("head_loss_x", HEAD_SHAPE)).requires_grad_(True)
    ex = {k: [torch.from_numpy(a) for a in v] for k, v in head_loss_example().items()}
    rets = head.loss(ex, head(x), {})
    total = sum(rets["loss"])
    total.backward()
    out = dict(loss=np.array([v.item() for v in rets["loss"]]), hm_loss=np.array([v.item() for v in rets["hm_loss"]]),
               loc_loss=np.array([v.item() for v in rets["loc_loss"]]),
               loc_loss_elem=np.stack([v.numpy() for v in rets["loc_loss_elem"]]),
               num_positive=np.array([v.item() for v in rets["num_positive"]]),
               gx=np.array([x.grad.sum(dtype=torch.float64).item(), x.grad.abs().sum(dtype=torch.float64).item()]),
               gw=np.array([head.shared_conv[0].weight.grad.abs().sum(dtype=torch.float64).item(),
                            head.tasks[1].hm[3].bias.grad.abs().sum(dtype=torch.float64).item()]))
    print({k: v.tolist() if v.size < 8 else v.shape for k, v in out.items()})
    save("centerhead_loss.npz", **out)


CONV_BWD_SHAPE, CONV_BWD_BATCH = [7, 20, 22], 2


def conv_bwd_case(subm):
    ind = detgen.clustered_voxels("bwd%d" % subm, CONV_BWD_BATCH, CONV_BWD_SHAPE, n_seeds=4, walk=120)
    ks, st, pd = ([3, 3, 3], [1, 1, 1], [1, 1, 1]) if subm else ([3, 3, 3], [2, 2, 2], [1, 1, 1])
    f = detgen.randn("bwd_f%d" % subm, (len(ind), 12))
    w = detgen.randn("bwd_w%d" % subm, (3, 3, 3, 12, 20), 0.2)
    return ind, ks, st, pd, f, w


def gen_conv_bwd():
    """indice_conv_backward_fp32 of the reference's compiled CPU code (oracle/_ref/sparse_conv_ext.so)."""
    out = {}
    for subm in (1, 0):
        ind, ks, st, pd, f, w = conv_bwd_case(subm)
        outids, pairs, num, _ = ref.get_indice_pairs(ind, CONV_BWD_BATCH, CONV_BWD_SHAPE, ks, st, pd, [1, 1, 1], subm)
        go = detgen.randn("bwd_g%d" % subm, (len(outids), 20))
        gi, gw = ref.indice_conv_backward(f, w, go, pairs, num, subm)
        order = np.lexsort(outids.T[::-1])                      # canonical out-voxel order
        out.update({"outids_%d" % subm: outids[order], "gi_%d" % subm: gi, "gw_%d" % subm: gw,
                    "order_%d" % subm: order.astype(np.int64)})
    save("conv_bwd.npz", **out)


def pool_points(name="dynvox"):
    """Points around a small grid: inside, outside, on the faces."""
    pts = detgen.rand(name, (4000, 4), -1.0, 1.0) * np.array([12.0, 9.0, 3.0, 1.0], np.float32)
    pts[:8, :3] = [[-10, -8, -2], [10, 8, 2], [9.999999, 0, 0], [0, 7.9999995, 0], [-10.000001, 0, 0], [0, 0, 1.9999999],
                   [0, -8, -2], [5, 5, 2.0000002]]
    return pts.astype(np.float32)


POOL_VS, POOL_RANGE = [0.25, 0.2, 0.5], [-10.0, -8.0, -2.0, 10.0, 8.0, 2.0]


def gen_pool():
    """indice_maxpool_fp32 / indice_maxpool_backward_fp32 / indice_conv_fp32(inverse) / dynamic_voxelize of the
    reference's compiled CPU code (oracle/_ref)."""
    out = {}
    ind, ks, st, pd, f, w = conv_bwd_case(0)                    # strided 3x3x3 geometry, 12 channels
    outids, pairs, num, _ = ref.get_indice_pairs(ind, CONV_BWD_BATCH, CONV_BWD_SHAPE, ks, st, pd, [1, 1, 1], 0)
    order = np.lexsort(outids.T[::-1])
    y = ref.indice_maxpool(f, pairs, num, len(outids))
    go = detgen.randn("pool_g", y.shape)
    fq = np.round(f * 2) / 2                                     # quantised copy: equal values inside one window
    yq = ref.indice_maxpool(fq, pairs, num, len(outids))
    out.update(outids=outids[order], order=order.astype(np.int64), y=y, gin=ref.indice_maxpool_backward(f, y, go, pairs, num),
               yq=yq, ginq=ref.indice_maxpool_backward(fq, yq, go, pairs, num))
    fo = detgen.randn("inv_f", (len(outids), 20))               # features on the strided conv's OUTPUT sites
    wi = detgen.randn("inv_w", (3, 3, 3, 20, 12), 0.2)
    out["inv"] = ref.indice_conv(fo, wi, pairs, num, len(ind), 0, inverse=True)
    out["dyn"] = ref.dynamic_voxelize(pool_points(), POOL_VS, POOL_RANGE)
    save("pool.npz", **out)


CONVT_CASES = (("k3s2p1", [3, 3, 3], [2, 2, 2], [1, 1, 1], [0, 0, 0]), ("k2s2", [2, 2, 2], [2, 2, 2], [0, 0, 0], [0, 0, 0]),
               ("k3s2p1op1", [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1]), ("k311s211", [3, 1, 1], [2, 1, 1], [0, 0, 0], [0, 0, 0]))


def convt_case():
    ind = detgen.clustered_voxels("convt", CONV_BWD_BATCH, CONV_BWD_SHAPE, n_seeds=3, walk=40)
    return ind, detgen.randn("convt_f", (len(ind), 16))


def gen_conv_transpose():
    """Transposed sparse convolution of the reference's compiled CPU code (oracle/_ref/sparse_conv_ext.so):
    get_indice_pairs_3d(transpose = 1) (ops.py:72-94, geometry.h:88-142,194-245) and indice_conv_fp32 on that rulebook
    (what SparseConvTranspose3d.forward runs, conv.py:114-204) for four geometries incl. output_padding."""
    ind, f = convt_case()
    out = {}
    for tag, ks, st, pd, op in CONVT_CASES:
        outids, pairs, num, oshape = ref.get_indice_pairs_transpose(ind, CONV_BWD_BATCH, CONV_BWD_SHAPE, ks, st, pd, [1, 1, 1], op)
        w = detgen.randn("convt_w_" + tag, tuple(ks) + (16, 16), 0.2)
        y = ref.indice_conv(f, w, pairs, num, len(outids), 0)
        order = np.lexsort(outids.T[::-1])                      # canonical out-voxel order
        out.update({"outids_" + tag: outids[order], "y_" + tag: y[order], "num_" + tag: num,
                    "oshape_" + tag: np.asarray(oshape, np.int32)})
    save("conv_transpose.npz", **out)


def gen_iou3d():
    """Rotated BEV IoU from the reference's own CPU path (oracle/_ref/iou3d_nms_cuda.so: boxes_iou_bev_cpu,
    CP/det3d/ops/iou3d_nms/src/iou3d_cpu.cpp:224-252) on detgen boxes, and the greedy keep list that the reference's
    host reduction (iou3d_nms.cpp:118-133) yields on THAT matrix for score-sorted boxes."""
    out = {}
    for tag, n, spread in (("dense", 192, 6.0), ("sparse", 300, 25.0)):
        a = detgen.bev_boxes("iou_a_" + tag, n, spread)
        b = detgen.bev_boxes("iou_b_" + tag, n - 17, spread, special=False)
        out["iou_" + tag] = ref.boxes_iou_bev_cpu(a, b)
        self_iou = ref.boxes_iou_bev_cpu(a, a)
        for thr in (0.2, 0.7):
            removed = np.zeros(n, bool)
            keep = []
            for i in range(n):
                if removed[i]:
                    continue
                keep.append(i)
                removed[i + 1:] |= self_iou[i, i + 1:] > thr
            out["keep_%s_%d" % (tag, int(thr * 100))] = np.asarray(keep, np.int64)
        out["self_iou_" + tag] = self_iou
    save("iou3d.npz", **out)


# ------------------------------------------------------------------ TransFusion fusion layer (point_fusion.ACTR)
TFF = dict(batch=2, n=350, ori_hw=(225, 400), in_hw=(112, 200), feat_hw=(28, 50), focal=316.0, yaw=2.7)


def _rot_z(deg):
    a = np.deg2rad(deg)
    return np.array([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]])


def _quat_from_matrix(R):
    """unit quaternion (w, x, y, z) of a rotation matrix (Shepperd's method)."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = [0.25 * s, (R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s]
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = [0.0, 0.0, 0.0, 0.0]
        q[0] = (R[k, j] - R[j, k]) / s
        q[1 + i] = 0.25 * s
        q[1 + j] = (R[j, i] + R[i, j]) / s
        q[1 + k] = (R[k, i] + R[i, k]) / s
    q = np.array(q)
    return q / np.linalg.norm(q)


class _Quaternion(object):
    """pyquaternion (0.9.x, absent here and unpinned by the reference) restated from its published definition:
    Quaternion([w, x, y, z]).rotation_matrix of the normalised quaternion."""

    def __init__(self, q):
        self.q = np.asarray(q, np.float64)

    @property
    def rotation_matrix(self):
        w, x, y, z = self.q / np.linalg.norm(self.q)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _view_points(points, view, normalize):
    """nuscenes-devkit (1.1.x, absent here and unpinned by the reference) `geometry_utils.view_points`, restated from
    its published definition: pad the intrinsic to 4x4, multiply, divide by depth."""
    viewpad = np.eye(4)
    viewpad[:view.shape[0], :view.shape[1]] = view
    nbr = points.shape[1]
    pts = np.concatenate((points, np.ones((1, nbr))))
    pts = np.dot(viewpad, pts)[:3, :]
    if normalize:
        pts = pts / pts[2:3, :].repeat(3, 0).reshape(3, nbr)
    return pts


def tff_calibration(sample):
    """Synthetic nuScenes records of one sample whose lidar -> camera chain (lidar sensor -> ego at the sweep -> global
    -> ego at the image -> camera sensor, point_fusion.py:586-607) composes to six pinhole cameras at 60 deg spacing.
    Returns (records for the fake DB, lidar2cam [6,4,4] float64 composed from the SAME records, intrinsics [6,3,3])."""
    cams = synth.nusc_cameras(image_hw=TFF["ori_hw"], focal=TFF["focal"], yaw_offset_deg=TFF["yaw"] + 0.9 * sample)
    R_ls, t_ls = _rot_z(1.3), np.array([0.94, 0.0, 1.84])                    # lidar sensor -> ego
    R_el, t_el = _rot_z(33.0 + sample), np.array([410.0 + sample, 1180.0, 0.0])          # ego (sweep time) -> global
    R_ec, t_ec = _rot_z(33.4 + sample), np.array([410.3 + sample, 1180.2, 0.0])          # ego (image time) -> global
    A = R_ec.T @ R_el @ R_ls
    avec = R_ec.T @ (R_el @ t_ls + t_el - t_ec)
    rec = {"calibrated_sensor": {"cs_lidar%d" % sample: dict(rotation=_quat_from_matrix(R_ls).tolist(), translation=t_ls.tolist())},
           "ego_pose": {"ep_lidar%d" % sample: dict(rotation=_quat_from_matrix(R_el).tolist(), translation=t_el.tolist())},
           "sample_data": {"sd_lidar%d" % sample: dict(calibrated_sensor_token="cs_lidar%d" % sample, ego_pose_token="ep_lidar%d" % sample)},
           "sample": {"s%d" % sample: dict(data={"LIDAR_TOP": "sd_lidar%d" % sample})}}
    l2c, Ks = [], []
    for name in synth.NUSC_CAMS:
        T, K = cams[name]
        T = T.astype(np.float64)
        R_cs = A @ T[:3, :3].T
        t_cs = avec - R_cs @ T[:3, 3]
        q = _quat_from_matrix(R_cs)
        rec["calibrated_sensor"]["cs%d_" % sample + name] = dict(rotation=q.tolist(), translation=t_cs.tolist(),
                                                      camera_intrinsic=K.astype(np.float64).tolist())
        rec["ego_pose"]["ep%d_" % sample + name] = dict(rotation=_quat_from_matrix(R_ec).tolist(), translation=t_ec.tolist())
        rec["sample_data"]["sd%d_" % sample + name] = dict(calibrated_sensor_token="cs%d_" % sample + name, ego_pose_token="ep%d_" % sample + name)
        rec["sample"]["s%d" % sample]["data"][name] = "sd%d_" % sample + name
        # compose from the records exactly as the reference walks them
        Rq = lambda r: _Quaternion(r).rotation_matrix              # noqa: E731
        Rls, Rel, Rec, Rcs = (Rq(rec["calibrated_sensor"]["cs_lidar%d" % sample]["rotation"]), Rq(rec["ego_pose"]["ep_lidar%d" % sample]["rotation"]),
                              Rq(rec["ego_pose"]["ep%d_" % sample + name]["rotation"]), Rq(q))
        Rt = Rcs.T @ Rec.T @ Rel @ Rls
        tt = Rcs.T @ (Rec.T @ (Rel @ t_ls + t_el - t_ec) - t_cs)
        M = np.eye(4)
        M[:3, :3], M[:3, 3] = Rt, tt
        l2c.append(M)
        Ks.append(K.astype(np.float64))
    return rec, np.stack(l2c), np.stack(Ks)


def tff_inputs():
    B, n = TFF["batch"], TFF["n"]
    rs = np.random.RandomState(77)
    pts = [np.concatenate([rs.uniform(-30, 30, (n, 2)), rs.uniform(-3, 1, (n, 1))], 1).astype(np.float32) for _ in range(B)]
    feats = detgen.randn("tff_feats", (B * n, 128))
    img = detgen.randn("tff_img", (B * 6, 256) + TFF["feat_hw"])
    return pts, feats, img


def tff_metas(aug):
    """img_metas of the two samples (reference keys) + the fake DB records + composed calibration for OUR layer."""
    ori, inp = TFF["ori_hw"], TFF["in_hw"]
    sf = [inp[1] / ori[1], inp[0] / ori[0], inp[1] / ori[1], inp[0] / ori[0]]
    metas, recs, l2cs, Ks = [], {}, [], []
    for b in range(TFF["batch"]):
        rec, l2c, K = tff_calibration(b)
        for tbl, d in rec.items():
            recs.setdefault(tbl, {}).update(d)
        m = dict(sample_idx="s%d" % b, filename=["samples/%s/n015__%s__%d.jpg" % (c, c, b) for c in synth.NUSC_CAMS],
                 ori_shape=(ori[0], ori[1], 3), img_shape=(inp[0], inp[1], 3), input_shape=(inp[0], inp[1]),
                 scale_factor=np.array(sf, np.float32), flip=False)
        if aug:
            ang = 0.3 - 0.5 * b
            c, s_ = np.cos(ang), np.sin(ang)
            # mmdet3d GlobalRotScaleTrans / RandomFlip3D records: rotation matrix applied as points @ R, scale, translation
            m.update(pcd_rotation=np.array([[c, -s_, 0], [s_, c, 0], [0, 0, 1]], np.float32).T, pcd_scale_factor=1.04 - 0.07 * b,
                     pcd_trans=np.array([0.4, -0.3, 0.1], np.float32) * (b + 1), pcd_horizontal_flip=(b == 0),
                     pcd_vertical_flip=(b == 1), transformation_3d_flow=['HF', 'VF', 'R', 'S', 'T'])
        metas.append(m)
        l2cs.append(l2c)
        Ks.append(K)
    return metas, recs, np.stack(l2cs), np.stack(Ks)


def import_reference_tf_fusion():
    R = "/root/reference/TransFusion/mmdet3d"
    for pkg, path in [("mmdet3d", R), ("mmdet3d.models", R + "/models"), ("mmdet3d.models.model_utils", R + "/models/model_utils"),
                      ("mmdet3d.models.model_utils.ops", R + "/models/model_utils/ops"), ("mmdet3d.ops", R + "/ops"),
                      ("mmdet3d.core", R + "/core"), ("mmdet3d.core.bbox", R + "/core/bbox"),
                      ("mmdet3d.models.fusion_layers", R + "/models/fusion_layers")]:
        _stub(pkg).__path__ = [path]
    _stub("cv2")
    tv = _stub("torchvision", __version__="0.25.0")
    tv.ops = _stub("torchvision.ops")
    tv.ops.misc = _stub("torchvision.ops.misc", _NewEmptyTensorOp=None)
    _stub("MultiScaleDeformableAttention")

    class ConvModule(torch.nn.Module):
        def __init__(s, *a, **k):
            super().__init__()

    _stub("mmcv")
    _stub("mmcv.cnn", ConvModule=ConvModule, xavier_init=lambda *a, **k: None)
    for n, a in [("mmdet3d.ops.gather_points.gather_points", "gather_points"),
                 ("mmdet3d.ops.furthest_point_sample.points_sampler", "Points_Sampler"),
                 ("mmdet3d.ops.group_points.group_points", "QueryAndGroup")]:
        _stub(n.rsplit(".", 1)[0])
        _stub(n, **{a: None})

    class _Registry(object):
        def register_module(self, *a, **k):
            return lambda cls: cls

    _stub("mmdet3d.models.registry", FUSION_LAYERS=_Registry())
    _stub("mmdet3d.core.bbox.structures", get_proj_mat_by_coord_type=lambda meta, coord: np.eye(4, dtype=np.float32))
    _stub("nuscenes")
    _stub("nuscenes.utils")
    _stub("nuscenes.utils.geometry_utils", view_points=_view_points)
    _stub("nuscenes.nuscenes", NuScenes=None)
    _stub("pyquaternion", Quaternion=_Quaternion)
    importlib.import_module("mmdet3d.core.points")                       # the reference's own LiDARPoints (pure torch)
    ct = importlib.import_module("mmdet3d.models.fusion_layers.coord_transform")
    sys.modules["mmdet3d.models.fusion_layers"].apply_3d_transformation = ct.apply_3d_transformation
    func = importlib.import_module("mmdet3d.models.model_utils.ops.functions.ms_deform_attn_func")

    class _F:
        apply = staticmethod(lambda v, s, l, loc, w, step: func.ms_deform_attn_core_pytorch(v, s, loc, w))

    importlib.import_module("mmdet3d.models.model_utils.ops.modules.ms_deform_attn").MSDeformAttnFunction = _F
    return importlib.import_module("mmdet3d.models.fusion_layers.point_fusion")


def gen_tf_fusion():
    """The reference's own fusion layer -- point_fusion.ACTR.forward with its get_2d_coor_multi / projection /
    split_param / agg_param (TF/mmdet3d/models/fusion_layers/point_fusion.py:342-643) -- on synthetic nuScenes records,
    without and with a 3-D augmentation flow to undo (coord_transform.py:6-94)."""
    pf = import_reference_tf_fusion()
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        layer = pf.ACTR(pfat_cfg=Cfg(ACTR_CFG)).eval()
    shapes = {k: tuple(v.shape) for k, v in layer.state_dict().items()}
    layer.load_state_dict({k: torch.from_numpy(v) for k, v in detgen.det_state_dict(shapes).items()})
    pts, feats, img = tff_inputs()
    out = {"param_names": np.array(sorted(shapes))}
    for tag, aug in (("plain", False), ("aug", True)):
        metas, recs, l2c, K = tff_metas(aug)

        class FakeNusc(object):
            def get(self, table, token):
                return recs[table][token]

        layer.nusc = FakeNusc()
        coords, coords_o = [], []
        # The reference runs on the GPU, where `points.cpu().numpy()` (point_fusion.py:585) is a COPY that its in-place
        # rotate / translate helpers may overwrite; on this CPU-only box the same expression aliases the input tensor
        # and camera n would see the points camera n-1 left behind.  Tensor.cpu() is made to copy for the duration.
        orig_cpu = torch.Tensor.cpu
        torch.Tensor.cpu = lambda self, *a, **k: orig_cpu(self, *a, **k).clone()
        for b, m in enumerate(metas):
            c2, c2o = pf.get_2d_coor_multi(img_meta=m, points=torch.from_numpy(pts[b]), proj_mat=torch.eye(4), coord_type='LIDAR',
                                           img_scale_factor=torch.from_numpy(m['scale_factor'][:2]), img_crop_offset=0,
                                           img_flip=False, img_pad_shape=m['input_shape'][:2], img_shape=m['img_shape'][:2],
                                           nusc=layer.nusc)
            coords.append(c2.numpy())
            coords_o.append(c2o.numpy())
        with torch.no_grad():
            fused = layer([torch.from_numpy(img)], [torch.from_numpy(p) for p in pts], torch.from_numpy(feats), metas, None)
        torch.Tensor.cpu = orig_cpu
        out[tag + "_coor_2d"] = np.concatenate(coords)
        out[tag + "_coor_2d_o"] = np.concatenate(coords_o)
        out[tag + "_fused"] = fused.numpy()
        out[tag + "_lidar2cam"] = l2c
        out[tag + "_intrinsic"] = K
        seen = np.concatenate(coords_o)[:, 1:].any(1)
        print(tag, "cameras used:", np.bincount(np.concatenate(coords)[:, 0].astype(int), minlength=6), "seen", int(seen.sum()))
    save("tf_fusion.npz", **out)


# ------------------------------------------------------------------ Voxel-RCNN fusion glue (VoxelBackBone8xFusion.point_fusion)
VRF = dict(batch=2, hw=(96, 320), n1=900, n4=260,
           lt=dict(npoint=64, radius=2.0, nsample=8, num_layers=2),
           actr=dict(fusion_method='sum', feature_modal='hybrid', num_bins=80, num_channels=[256], query_num_feat=64,
                     num_enc_layers=4, max_num_ne_voxel=20000, pos_encode_method='depth'),
           hybrid=dict(attn_layer='BiGateSum1D_2', q_method='sum', q_rep_place=['weight']))


def vrf_calib(b):
    """KITTI-style calibration of sample b: P2 [3,4], R0 [3,3], Tr_velo2cam [3,4] (float32 as the devkit files are read)."""
    H, W = VRF["hw"]
    P2 = np.array([[180.0 + 3 * b, 0, W / 2 + 1.5, 11.0], [0, 180.0 + 3 * b, H / 2 - 0.7, 0.3], [0, 0, 1, 0.002]], np.float32)
    a = np.deg2rad(0.4 + 0.3 * b)
    R0 = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
    V2C = np.array([[0.003, -1, 0.01, 0.01], [0.012, 0.008, -1, -0.08], [1, 0.004, 0.011, -0.27]], np.float32)
    return P2, R0, V2C


def vrf_inputs():
    """Index sets at stride 1 (conv1, 16 ch) and stride 8 (conv4, 64 ch) on the KITTI grid [41, 1600, 1408], the two
    camera feature maps, and per-sample augmentation records (noise_scale, noise_rot, flip_x)."""
    B, (H, W) = VRF["batch"], VRF["hw"]
    rs = np.random.RandomState(11)

    def voxels(n, stride):
        out = []
        for b in range(B):
            # x in 4..45 m, |y| < 12 m, z -2.5 .. 0.5: inside the camera frustum mostly, some outside
            x = rs.uniform(4, 45, n), rs.uniform(-12, 12, n), rs.uniform(-2.6, 0.6, n)
            zi = np.floor((x[2] + 3) / (0.1 * stride)).astype(np.int32)
            yi = np.floor((x[1] + 40) / (0.05 * stride)).astype(np.int32)
            xi = np.floor(x[0] / (0.05 * stride)).astype(np.int32)
            ind = np.unique(np.stack([np.full(n, b, np.int32), zi, yi, xi], 1), axis=0)
            out.append(ind)
        return np.concatenate(out)
    ind1, ind4 = voxels(VRF["n1"], 1), voxels(VRF["n4"], 8)
    f1 = detgen.randn("vrf_f1", (len(ind1), 16))
    f4 = detgen.randn("vrf_f4", (len(ind4), 64))
    mvx = detgen.randn("vrf_mvx", (B, 16, H // 4, W // 4))
    img = detgen.randn("vrf_img", (B, 256, H // 4, W // 4))
    aug = dict(noise_scale=np.array([1.03, 0.96], np.float32), noise_rot=np.array([0.21, -0.33], np.float32),
               flip_x=np.array([True, False]))
    return ind1, f1, ind4, f4, mvx, img, aug


def import_reference_vr_backbone():
    R = "/root/reference/VoxelRCNN/pcdet"
    for pkg, path in [("pcdet", R), ("pcdet.models", R + "/models"), ("pcdet.models.model_utils", R + "/models/model_utils"),
                      ("pcdet.models.model_utils.ops", R + "/models/model_utils/ops"), ("pcdet.ops", R + "/ops"),
                      ("pcdet.utils", R + "/utils"), ("pcdet.models.backbones_3d", R + "/models/backbones_3d"),
                      ("pcdet.models.backbones_3d.SemanticSeg", R + "/models/backbones_3d/SemanticSeg")]:
        _stub(pkg).__path__ = [path]
    _stub("cv2")
    _stub("SharedArray")
    tv = _stub("torchvision", __version__="0.25.0")
    tv.ops = _stub("torchvision.ops")
    tv.ops.misc = _stub("torchvision.ops.misc", _NewEmptyTensorOp=None)
    _stub("MultiScaleDeformableAttention")
    _stub("mmcv")

    class ConvModule(torch.nn.Module):
        """mmcv.cnn.ConvModule (absent) restated for the two configurations pointformer.py uses."""

        def __init__(self, cin, cout, k, norm_cfg=None, act_cfg=dict(type="ReLU")):
            super().__init__()
            self.conv = torch.nn.Conv2d(cin, cout, k, bias=norm_cfg is None)
            if norm_cfg is not None:
                self.bn = torch.nn.BatchNorm2d(cout)
            if act_cfg is not None:
                self.activate = torch.nn.ReLU(inplace=True)

        def forward(self, x):
            x = self.conv(x)
            if hasattr(self, "bn"):
                x = self.bn(x)
            if hasattr(self, "activate"):
                x = self.activate(x)
            return x

    _stub("mmcv.cnn", ConvModule=ConvModule)
    from oracle import oracle as orc

    class Sampler(torch.nn.Module):
        def __init__(self, num_point, mods):
            super().__init__()
            self.m = num_point[0]

        def forward(self, xyz, feats):
            return torch.from_numpy(orc.furthest_point_sample(xyz.numpy(), self.m))

    class Grouper(torch.nn.Module):
        def __init__(self, radius, nsample, **kw):
            super().__init__()
            self.r, self.ns = radius, nsample

        def forward(self, xyz, new_xyz, feats):
            idx = orc.ball_query(0.0, self.r, self.ns, xyz.numpy(), new_xyz.numpy())
            gx = orc.group_points(xyz.transpose(1, 2).contiguous().numpy(), idx)
            gf = orc.group_points(feats.numpy(), idx)
            return torch.from_numpy(gf), torch.from_numpy(gx), torch.from_numpy(idx)

    # the reference's four CUDA-only index ops are bound to the oracle (pinned to the reference's own unit-test literals
    # by gen_pointops), exactly as gen_local_transformer does
    _stub("pcdet.ops.gather_points")
    _stub("pcdet.ops.gather_points.gather_points",
          gather_points=lambda f, i: torch.from_numpy(orc.gather_points(f.numpy(), i.numpy())))
    _stub("pcdet.ops.furthest_point_sample")
    _stub("pcdet.ops.furthest_point_sample.points_sampler", Points_Sampler=Sampler)
    _stub("pcdet.ops.group_points")
    _stub("pcdet.ops.group_points.group_points", QueryAndGroup=Grouper)
    # import-only stubs of modules the fusion glue never touches

    class _SparseT(object):
        def __init__(self, features, indices, spatial_shape, batch_size):
            self.features, self.indices, self.spatial_shape, self.batch_size = features, indices, spatial_shape, batch_size

    sp = types.SimpleNamespace(SparseModule=torch.nn.Module, SparseConvTensor=_SparseT, SparseSequential=torch.nn.Sequential)
    _stub("pcdet.utils.spconv_utils", spconv=sp, replace_feature=lambda t, f: _SparseT(f, t.indices, t.spatial_shape, t.batch_size))
    _stub("pcdet.models.backbones_3d.SemanticSeg.pyramid_ffn", PyramidFeat2D=None)
    _stub("pcdet.models.backbones_3d.SemanticSeg.aux_seg_loss", AuxConsistencyLoss=None)
    _stub("pcdet.models.dense_heads", __all__={})
    _stub("pcdet.models.model_utils.attention", __all__={})
    func = importlib.import_module("pcdet.models.model_utils.ops.functions.ms_deform_attn_func")

    class _F:
        apply = staticmethod(lambda v, s, l, loc, w, step: func.ms_deform_attn_core_pytorch(v, s, loc, w))

    importlib.import_module("pcdet.models.model_utils.ops.modules.ms_deform_attn").MSDeformAttnFunction = _F
    pf = importlib.import_module("pcdet.models.model_utils.pointformer")
    _fw = pf.TransformerEncoderLayerPreNorm.forward
    pf.TransformerEncoderLayerPreNorm.forward = lambda self, src, src_mask=None, src_key_padding_mask=None, **kw: \
        _fw(self, src, src_mask, src_key_padding_mask)
    bb = importlib.import_module("pcdet.models.backbones_3d.spconv_backbone")
    cal = importlib.import_module("pcdet.utils.calibration_kitti")
    actr = importlib.import_module("pcdet.models.model_utils.actr")
    return bb, cal, actr, sp


def gen_vr_fusion():
    """The reference's own VoxelBackBone8xFusion.point_fusion (VR/pcdet/models/backbones_3d/spconv_backbone.py:650-827)
    called as a plain function on a stand-in `self` that carries exactly the attributes it reads, with the reference's
    KITTI `Calibration.lidar_to_img`, its `rotate_points_along_z`, and the reference's ACTRv2 (d_model 64, 4 encoder
    layers, LocalTransformer per layer) built by its own `build`: (a) the MVX nearest-pixel sum at stride 1, (b) the
    ACTRv2 dual-query fusion at stride 8, each without and with augmentation records to undo."""
    bb, cal, actr_mod, sp = import_reference_vr_backbone()
    torch.set_num_threads(1)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        actr = actr_mod.build(Cfg(VRF["actr"]), model_name="ACTRv2", lt_cfg=Cfg(VRF["lt"]), hybrid_cfg=Cfg(VRF["hybrid"])).eval()
    shapes = {k: tuple(v.shape) for k, v in actr.state_dict().items()}
    actr.load_state_dict({k: torch.from_numpy(v) for k, v in detgen.det_state_dict(shapes).items()})
    ind1, f1, ind4, f4, mvx, img, aug = vrf_inputs()
    B, (H, W) = VRF["batch"], VRF["hw"]
    calibs, l2i = [], []
    for b in range(B):
        P2, R0, V2C = vrf_calib(b)
        calibs.append(cal.Calibration(dict(P2=P2, R0=R0, Tr_velo2cam=V2C)))
        # the same chain in float64 for OUR layer (VoxelBackBone8xFusion.lidar2img_from_kitti): rows 0, 1 of
        # P2 @ [R0 0; 0 1] @ [V2C; 0 0 0 1], row 2 = the rectified-camera depth row -- the devkit divides by the rect
        # depth, not by the homogeneous coordinate (calibration_kitti.py:80-82)
        R0e, Ve = np.eye(4), np.eye(4)
        R0e[:3, :3], Ve[:3, :4] = R0, V2C
        M = P2.astype(np.float64) @ R0e @ Ve
        M[2] = (R0e @ Ve)[2]
        l2i.append(M)
    me = types.SimpleNamespace(voxel_size=torch.tensor([0.1, 0.05, 0.05]), point_cloud_range=torch.tensor([-3., -40., 0., 1., 40., 70.4]),
                               inv_idx=torch.tensor([2, 1, 0]), max_num_nev=VRF["actr"]["max_num_ne_voxel"], actr=actr,
                               attention=False)
    out = dict(param_names=np.array(sorted(shapes)), lidar2img=np.stack(l2i))
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self                 # the glue calls .cuda() on fresh tensors; no GPU here
    try:
        for tag, with_aug in (("plain", False), ("aug", True)):
            bd = dict(calib=calibs, batch_size=B, images=torch.zeros(B, 3, H, W))
            if with_aug:
                bd.update(noise_scale=torch.from_numpy(aug["noise_scale"]), noise_rot=torch.from_numpy(aug["noise_rot"]),
                          flip_x=torch.from_numpy(aug["flip_x"]))
            x1 = sp.SparseConvTensor(torch.from_numpy(f1.copy()), torch.from_numpy(ind1.copy()), [41, 1600, 1408], B)
            with torch.no_grad():
                y1 = bb.VoxelBackBone8xFusion.point_fusion(me, [x1], bd, {"mvx_layer1_feat2d": torch.from_numpy(mvx)}, "MVX", 1)
            x4 = sp.SparseConvTensor(torch.from_numpy(f4.copy()), torch.from_numpy(ind4.copy()), [5, 200, 176], B)
            stages, hooks = {}, []
            enc = actr.transformer.encoder
            for nm, mod in (("lt0", enc.lidar_attns[0]), ("layer0", enc.layers[0]), ("lt1", enc.lidar_attns[1])):
                hooks.append(mod.register_forward_hook(
                    lambda m_, i_, o_, nm=nm: stages.__setitem__(nm, (o_[0] if isinstance(o_, tuple) else o_).detach().numpy().copy())))
            def grab(m_, args, kwargs):
                stages["in_ref"] = args[2].detach().numpy().copy()
                stages["in_qpos"] = kwargs["q_pos"].detach().numpy().copy()
                stages["in_qi"] = kwargs["q_i_feat"].detach().numpy().copy()
            hooks.append(enc.layers[0].register_forward_pre_hook(grab, with_kwargs=True))
            with torch.no_grad():
                y4 = bb.VoxelBackBone8xFusion.point_fusion(me, [x4], bd, {"layer1_feat2d": torch.from_numpy(img)}, "ACTRv2", 8)
            for h_ in hooks:
                h_.remove()
            for nm, v in stages.items():               # intermediate LiDAR queries [B, n_max, 64] (localise a deviation)
                out[tag + "_stage_" + nm] = v
            out[tag + "_mvx"] = y1.features.numpy()
            out[tag + "_actr"] = y4.features.numpy()
            print(tag, "MVX rows changed:", int((np.abs(y1.features.numpy() - f1).max(1) > 0).sum()), "of", len(f1),
                  "| ACTR |delta| max %.3f" % np.abs(y4.features.numpy() - f4).max())
    finally:
        torch.Tensor.cuda = orig_cuda
    save("vr_fusion.npz", **out)


def gen_vr_gate():
    """The reference's own `BasicGate` (VR/pcdet/models/model_utils/attention.py:88-177, the image gate of
    `voxel_rcnn_car_mm_mvx+actrv2_hybrid_ifat.yaml`; called at spconv_backbone.py:797-800 with x_list = [x_conv2, x_conv3,
    x_conv4] and one image level): stride-2 voxels (32 channels) projected with its KITTI `Calibration.lidar_to_img`, scattered
    by its `pts2img`, two 3x3 convolutions, sigmoid, product with the image features -- without and with augmentation records.
    Its convolution stack lives in a plain Python list (not registered: no state_dict keys); the values are stored here."""
    bb, cal, actr_mod, sp = import_reference_vr_backbone()
    import importlib
    sys.modules.pop("pcdet.models.model_utils.attention", None)          # the import harness above stubs it for the backbone
    att_mod = importlib.import_module("pcdet.models.model_utils.attention")     # the reference's own file
    torch.set_num_threads(1)
    B, (H, W) = VRF["batch"], VRF["hw"]
    rs = np.random.RandomState(23)
    inds = []
    for b in range(B):
        n = 1500
        x = rs.uniform(2, 60, n), rs.uniform(-25, 25, n), rs.uniform(-2.8, 0.8, n)      # part of them outside the image
        zi = np.floor((x[2] + 3) / 0.2).astype(np.int32)
        yi = np.floor((x[1] + 40) / 0.1).astype(np.int32)
        xi = np.floor(x[0] / 0.1).astype(np.int32)
        inds.append(np.unique(np.stack([np.full(n, b, np.int32), zi, yi, xi], 1), axis=0))
    ind2 = np.concatenate(inds)
    f2 = detgen.randn("vrg_f2", (len(ind2), 32))
    img = detgen.randn("vrg_img", (B, 256, H // 4, W // 4))
    gate = att_mod.BasicGate(img_channel_list=[256], pts_channel_list=[32], sparse_shape=[41, 1600, 1408],
                             voxel_size=torch.tensor([0.1, 0.05, 0.05]), point_cloud_range=torch.tensor([-3., -40., 0., 1., 40., 70.4]),
                             inv_idx=torch.tensor([2, 1, 0]), pts_idx=[0]).eval()
    stack = gate.spatial_basic_list[0].eval()
    shapes = {k: tuple(v.shape) for k, v in stack.state_dict().items()}
    sd = detgen.det_state_dict(shapes)
    stack.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    calibs, l2i = [], []
    for b in range(B):
        P2, R0, V2C = vrf_calib(b)
        calibs.append(cal.Calibration(dict(P2=P2, R0=R0, Tr_velo2cam=V2C)))
        R0e, Ve = np.eye(4), np.eye(4)
        R0e[:3, :3], Ve[:3, :4] = R0, V2C
        M = P2.astype(np.float64) @ R0e @ Ve
        M[2] = (R0e @ Ve)[2]
        l2i.append(M)
    aug = dict(noise_scale=np.array([1.03, 0.96], np.float32), noise_rot=np.array([0.21, -0.33], np.float32),
               flip_x=np.array([True, False]))
    out = dict(ind2=ind2, lidar2img=np.stack(l2i), **{"stack_" + k: v for k, v in sd.items()})
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for tag, with_aug in (("plain", False), ("aug", True)):
            bd = dict(calib=calibs, batch_size=B, images=torch.zeros(B, 3, H, W))
            if with_aug:
                bd.update(noise_scale=torch.from_numpy(aug["noise_scale"]), noise_rot=torch.from_numpy(aug["noise_rot"]),
                          flip_x=torch.from_numpy(aug["flip_x"]))
            x2 = sp.SparseConvTensor(torch.from_numpy(f2.copy()), torch.from_numpy(ind2.copy()), [21, 800, 704], B)
            with torch.no_grad():
                y = gate(x_rgb=[torch.from_numpy(img)], x_list=[x2], batch_dict=bd)
            out[tag + "_gated"] = y[0].numpy()[:, :4].copy()         # (the gate is one map per pixel: four channels pin it)
            a = (y[0].numpy() / np.where(img == 0, 1, img))[:, 0]
            print(tag, "gate range %.3f .. %.3f, pixels != sigmoid(bias-only) %d" % (a.min(), a.max(), int((np.abs(a - np.median(a)) > 1e-6).sum())))
    finally:
        torch.Tensor.cuda = orig_cuda
    save("vr_gate.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["voxelize", "rulebook", "msda", "actr", "fusion", "pointops", "lt", "iou3d", "centerhead", "tfhead", "tfloss", "headloss", "conv_bwd", "pool", "conv_transpose"]
    if "iou3d" in which:
        gen_iou3d()
    if "conv_bwd" in which:
        gen_conv_bwd()
    if "pool" in which:
        gen_pool()
    if "conv_transpose" in which:
        gen_conv_transpose()
    if "centerhead" in which:
        gen_centerhead()
    if "tfhead" in which:
        gen_transfusion_head()
    if "tfloss" in which:
        gen_transfusion_head_loss()
    if "headloss" in which:
        gen_centerhead_loss()
    if "voxelize" in which:
        gen_voxelize()
    if "rulebook" in which:
        gen_rulebook_conv()
    if "msda" in which or "actr" in which:
        actr, func = import_reference_actr()
        if "msda" in which:
            gen_msda(func)
            gen_msda_bwd(func)
        if "actr" in which:
            gen_actr(actr)
    if "pointops" in which:
        gen_pointops()
    if "lt" in which:
        gen_local_transformer()
    if "tf_fusion" in which:
        gen_tf_fusion()
    if "vr_fusion" in which:
        gen_vr_fusion()
    if "vr_gate" in which:
        gen_vr_gate()
    if "fusion" in which:
        if "det3d" not in sys.modules:
            import_reference_actr()
        gen_fusion()
