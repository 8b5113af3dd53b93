"""Boundary, CPU part: the extension shims carry the reference's pybind names
(TF/mmdet3d/ops/spconv/src/all.cc:21-51, voxel/src/voxelization.cpp:7-11, CP/.../ops/src/vision.cpp:13-16, the four
point-op modules, iou3d), refuse CPU tensors with a RuntimeError (no CPU fallback), install under the import paths the
reference's wrappers use -- and `registry.late_register()` really puts our classes into the frameworks' registries
(fake mmcv / mmdet3d / mmdet / det3d / pcdet modules stand in for the absent packages)."""
import sys
import types

import pytest

torch = pytest.importorskip("torch")

_FRAMEWORKS = ("mmcv", "mmdet", "mmdet3d", "det3d", "pcdet", "MultiScaleDeformableAttention")

PYBIND = {
    "sparse_conv_ext": ["get_indice_pairs_2d", "get_indice_pairs_3d", "get_indice_pairs_4d", "get_indice_pairs_grid_2d",
                        "get_indice_pairs_grid_3d", "indice_conv_fp32", "indice_conv_backward_fp32", "indice_conv_half",
                        "indice_conv_backward_half", "fused_indice_conv_fp32", "fused_indice_conv_half", "indice_maxpool_fp32",
                        "indice_maxpool_backward_fp32", "indice_maxpool_half", "indice_maxpool_backward_half"],
    "voxel_layer": ["hard_voxelize", "dynamic_voxelize", "dynamic_point_to_voxel_forward", "dynamic_point_to_voxel_backward"],
    "MultiScaleDeformableAttention": ["ms_deform_attn_forward", "ms_deform_attn_backward"],
    "furthest_point_sample_ext": ["furthest_point_sampling_wrapper", "furthest_point_sampling_with_dist_wrapper"],
    "ball_query_ext": ["ball_query_wrapper"],
    "group_points_ext": ["forward", "backward"],
    "gather_points_ext": ["gather_points_wrapper", "gather_points_grad_wrapper"],
    "iou3d_cuda": ["boxes_overlap_bev_gpu", "boxes_iou_bev_gpu", "nms_gpu", "nms_normal_gpu"],
}


def test_shims_export_the_pybind_names_and_refuse_cpu_tensors():
    from dualfusion import ext
    assert sorted(ext.NAMES) == sorted(PYBIND)
    for mod, names in PYBIND.items():
        m = ext.load(mod)
        for n in names:
            assert callable(getattr(m, n)), (mod, n)
    pts = torch.zeros(10, 5)
    vl = ext.load("voxel_layer")
    with pytest.raises(RuntimeError):
        vl.hard_voxelize(pts, torch.zeros(4, 3, 5), torch.zeros(4, 3, dtype=torch.int32), torch.zeros(4, dtype=torch.int32),
                         [0.1, 0.1, 0.1], [0, 0, 0, 1, 1, 1], 3, 4)
    with pytest.raises(RuntimeError):
        vl.dynamic_voxelize(pts, torch.zeros(10, 3, dtype=torch.int32), [0.1, 0.1, 0.1], [0, 0, 0, 1, 1, 1])
    with pytest.raises(RuntimeError):
        vl.dynamic_point_to_voxel_forward(pts, pts, "mean")
    sp = ext.load("sparse_conv_ext")
    ind = torch.zeros(3, 4, dtype=torch.int32)
    with pytest.raises(RuntimeError):
        sp.get_indice_pairs_3d(ind, 1, [4, 4, 4], [4, 4, 4], [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], [0, 0, 0], 1, 0)
    with pytest.raises(RuntimeError):
        sp.get_indice_pairs_4d(ind, 1)
    with pytest.raises(RuntimeError):
        sp.indice_conv_fp32(torch.zeros(3, 4), torch.zeros(3, 3, 3, 4, 8), torch.zeros(27, 2, 3, dtype=torch.int32),
                            torch.zeros(27, dtype=torch.int32), 3, 0, 1)
    ms = ext.load("MultiScaleDeformableAttention")
    with pytest.raises(RuntimeError):
        ms.ms_deform_attn_forward(torch.zeros(1, 4, 2, 2), torch.tensor([[2, 2]]), torch.tensor([0]), torch.zeros(1, 3, 2, 1, 2, 2),
                                  torch.zeros(1, 3, 2, 1, 2), 64)
    with pytest.raises(RuntimeError):
        ext.load("furthest_point_sample_ext").furthest_point_sampling_wrapper(1, 8, 2, torch.zeros(1, 8, 3), torch.zeros(1, 8),
                                                                              torch.zeros(1, 2, dtype=torch.int32))
    with pytest.raises(RuntimeError):
        ext.load("ball_query_ext").ball_query_wrapper(1, 8, 2, 0.0, 1.0, 4, torch.zeros(1, 2, 3), torch.zeros(1, 8, 3),
                                                      torch.zeros(1, 2, 4, dtype=torch.int32))
    with pytest.raises(RuntimeError):
        ext.load("group_points_ext").forward(1, 2, 8, 2, 4, torch.zeros(1, 2, 8), torch.zeros(1, 2, 4, dtype=torch.int32),
                                             torch.zeros(1, 2, 2, 4))
    with pytest.raises(RuntimeError):
        ext.load("gather_points_ext").gather_points_wrapper(1, 2, 8, 2, torch.zeros(1, 2, 8), torch.zeros(1, 2, dtype=torch.int32),
                                                            torch.zeros(1, 2, 2))
    with pytest.raises(RuntimeError):
        ext.load("iou3d_cuda").boxes_overlap_bev_gpu(torch.zeros(2, 5), torch.zeros(3, 5), torch.zeros(2, 3))


def test_install_binds_the_reference_import_paths():
    from dualfusion import ext
    saved = dict(sys.modules)
    try:
        for k in [k for k in sys.modules if k.split(".")[0] in ("mmdet3d", "det3d", "MultiScaleDeformableAttention")]:
            del sys.modules[k]
        pkg = types.ModuleType("mmdet3d.ops.spconv")          # an already imported parent package gets the attribute
        sys.modules["mmdet3d.ops.spconv"] = pkg
        done = ext.install()
        assert "mmdet3d.ops.spconv.sparse_conv_ext" in done and "MultiScaleDeformableAttention" in done
        import MultiScaleDeformableAttention as MSDA          # ms_deform_attn_func.py:18
        assert MSDA.ms_deform_attn_forward is ext.load("MultiScaleDeformableAttention").ms_deform_attn_forward
        assert pkg.sparse_conv_ext is ext.load("sparse_conv_ext")
        assert sys.modules["mmdet3d.ops.voxel.voxel_layer"].hard_voxelize is ext.load("voxel_layer").hard_voxelize
        assert sys.modules["det3d.ops.ball_query.ball_query_ext"] is ext.load("ball_query_ext")
        sentinel = types.ModuleType("compiled")               # a compiled module that is already loaded wins ...
        sys.modules["mmdet3d.ops.iou3d.iou3d_cuda"] = sentinel
        assert "mmdet3d.ops.iou3d.iou3d_cuda" not in ext.install()
        assert sys.modules["mmdet3d.ops.iou3d.iou3d_cuda"] is sentinel
        assert "mmdet3d.ops.iou3d.iou3d_cuda" in ext.install(overwrite=True)      # ... unless asked otherwise
    finally:
        for k in list(sys.modules):                               # only the stand-ins go; product modules stay imported
            if k not in saved and k.split(".")[0] in _FRAMEWORKS:
                del sys.modules[k]
        sys.modules.update({k: v for k, v in saved.items() if k.split(".")[0] in _FRAMEWORKS})


class _MMRegistry(object):
    """mmcv.utils.Registry's registration surface (register_module(name=, force=, module=))."""

    def __init__(self):
        self.module_dict = {}

    def register_module(self, name=None, force=False, module=None):
        if module is None:
            raise TypeError("used as a function here")
        if name in self.module_dict and not force:
            raise KeyError(name)
        self.module_dict[name] = module
        return module


class _Det3dRegistry(object):
    def __init__(self):
        self._module_dict = {}


def test_late_register_fills_the_frameworks_registries():
    """CP / TF / VR configs resolve `type=` / `NAME:` strings through the frameworks' OWN registries: after
    late_register() those hold the MI355X classes (TF/mmdet3d/models/builder.py, CP/det3d/models/registry.py,
    VR/pcdet/models/backbones_3d/__init__.py `__all__`)."""
    import dualfusion.backbones, dualfusion.fusion, dualfusion.heads, dualfusion.necks, dualfusion.spconv, dualfusion.voxel  # noqa: F401,E401
    import dualfusion.transfusion_head  # noqa: F401
    from dualfusion import registry as R
    saved = dict(sys.modules)
    try:
        conv = _MMRegistry()
        conv.module_dict["SubMConv3d"] = object                      # the framework's own (compiled-op) class is replaced
        fl, me, ve = _MMRegistry(), _MMRegistry(), _MMRegistry()
        mb, mh, mn = _MMRegistry(), _MMRegistry(), _MMRegistry()
        sys.modules["mmcv"] = types.ModuleType("mmcv")
        sys.modules["mmcv.cnn"] = types.ModuleType("mmcv.cnn")
        sys.modules["mmcv.cnn"].CONV_LAYERS = conv
        for n in ("mmdet3d", "mmdet3d.models"):
            sys.modules[n] = types.ModuleType(n)
        b = types.ModuleType("mmdet3d.models.builder")
        b.FUSION_LAYERS, b.MIDDLE_ENCODERS, b.VOXEL_ENCODERS = fl, me, ve
        sys.modules["mmdet3d.models.builder"] = b
        sys.modules["mmdet"] = types.ModuleType("mmdet")
        m = types.ModuleType("mmdet.models")
        m.BACKBONES, m.HEADS, m.NECKS = mb, mh, mn
        md = _MMRegistry()
        m.DETECTORS = md
        sys.modules["mmdet.models"] = m
        d3 = types.ModuleType("det3d.models.registry")
        for n in ("READERS", "BACKBONES", "FUSION", "NECKS", "HEADS"):
            setattr(d3, n, _Det3dRegistry())
        sys.modules["det3d"] = types.ModuleType("det3d")
        sys.modules["det3d.models"] = types.ModuleType("det3d.models")
        sys.modules["det3d.models"].registry = d3
        sys.modules["det3d.models.registry"] = d3
        p3 = types.ModuleType("pcdet.models.backbones_3d")
        p3.__all__ = {"VoxelBackBone8x": object}
        sys.modules["pcdet"] = types.ModuleType("pcdet")
        sys.modules["pcdet.models"] = types.ModuleType("pcdet.models")
        sys.modules["pcdet.models.backbones_3d"] = p3
        sys.modules["pcdet.models"].backbones_3d = p3
        done = R.late_register()
        assert done == ["mmcv.cnn.CONV_LAYERS", "mmdet3d", "mmdet", "det3d", "pcdet"], done
        from dualfusion.spconv import SparseConv3d, SubMConv3d
        assert conv.module_dict["SubMConv3d"] is SubMConv3d and conv.module_dict["SparseConv3d"] is SparseConv3d
        assert me.module_dict["SparseEncoderFusion"] is dualfusion.backbones.SparseEncoderFusion
        assert "HardSimpleVFE" in ve.module_dict and "ACTR" in fl.module_dict
        assert mh.module_dict["TransFusionHead"] is dualfusion.transfusion_head.TransFusionHead
        assert "SECOND" in mb.module_dict and "SECONDFPN" in mn.module_dict
        assert md.module_dict["TransFusionDetector"] is dualfusion.transfusion.TransFusionDetector      # round 6
        assert d3.BACKBONES._module_dict["SpMiddleResNetFHDFusion"] is dualfusion.backbones.SpMiddleResNetFHDFusion
        assert d3.HEADS._module_dict["CenterHead"] is dualfusion.heads.CenterHead and "RPN" in d3.NECKS._module_dict
        assert "VoxelFeatureExtractorV3" in d3.READERS._module_dict and len(d3.FUSION._module_dict) > 0
        assert p3.__all__["VoxelBackBone8xFusion"] is dualfusion.backbones.VoxelBackBone8xFusion
        # a reference-style config resolves through the framework's registry to our class
        cfg = dict(type="SubMConv3d", in_channels=4, out_channels=8, kernel_size=3, indice_key="subm1")
        args = dict(cfg)
        layer = conv.module_dict[args.pop("type")](**args)
        assert isinstance(layer, SubMConv3d) and tuple(layer.weight.shape) == (3, 3, 3, 4, 8)
    finally:
        for k in list(sys.modules):                               # only the stand-ins go; product modules stay imported
            if k not in saved and k.split(".")[0] in _FRAMEWORKS:
                del sys.modules[k]
        sys.modules.update({k: v for k, v in saved.items() if k.split(".")[0] in _FRAMEWORKS})
    assert R.late_register() == []                                   # nothing importable: nothing claimed


def test_tall_skinny_linear_autograd_functions_match_torch():
    """dualfusion.ops.linear_rows_autograd / channel_first_linear (training: the weight gradient over ~10^5 rows as a batched
    product over row chunks) against torch's own autograd in float64 on the CPU."""
    import torch
    from dualfusion import ops
    torch.manual_seed(0)
    x = torch.randn(3, 40, 1602 * 8, dtype=torch.float64)
    w = torch.randn(24, 40, dtype=torch.float64, requires_grad=True)
    y = ops._ChannelFirstLinear.apply(x, w)
    g = torch.randn_like(y)
    y.backward(g)
    w2 = w.detach().clone().requires_grad_(True)
    y2 = torch.matmul(w2, x)
    y2.backward(g)
    assert torch.allclose(y, y2) and torch.allclose(w.grad, w2.grad, rtol=1e-10, atol=1e-10)
    xr = torch.randn(2, 1602 * 9, 16, dtype=torch.float64, requires_grad=True)
    wl = torch.randn(8, 16, dtype=torch.float64, requires_grad=True)
    b = torch.randn(8, dtype=torch.float64, requires_grad=True)
    y = ops.linear_rows_autograd(xr, wl, b)
    g = torch.randn_like(y)
    y.backward(g)
    ref = [t.detach().clone().requires_grad_(True) for t in (xr, wl, b)]
    torch.nn.functional.linear(*ref).backward(g)
    for a, r in zip((xr, wl, b), ref):
        assert torch.allclose(a.grad, r.grad, rtol=1e-10, atol=1e-10)
