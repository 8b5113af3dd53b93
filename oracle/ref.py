"""oracle/ref.py -- loader for the REAL reference CPU builds in oracle/_ref/ (test infrastructure).

The shared objects are produced by oracle/build_ref.py from the sources under
/root/reference.  `sparse_conv_ext.so` is linked without the reference's GPU rulebook
functors (src/indice_cuda.cu does not compile under hipcc); it is therefore loaded
with RTLD_LAZY so that those never-called symbols stay unresolved.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = os.path.join(_HERE, "_ref")
_mods = {}


def available(name="sparse_conv_ext"):
    return os.path.exists(os.path.join(_REF, name + ".so"))


def load(name):
    """name in {'sparse_conv_ext', 'voxel_layer', 'iou3d_nms_cuda', 'iou3d_cuda'} -> the reference's pybind module."""
    if name in _mods:
        return _mods[name]
    import torch  # noqa: F401  (libtorch must be loaded first)
    path = os.path.join(_REF, name + ".so")
    if not os.path.exists(path):
        raise FileNotFoundError(path + " (run oracle/build_ref.py where /root/reference exists)")
    old = sys.getdlopenflags()
    sys.setdlopenflags(os.RTLD_LAZY)
    try:
        spec = importlib.util.spec_from_file_location(name, path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        sys.setdlopenflags(old)
    _mods[name] = m
    return m


# ---- thin numpy-facing wrappers with the reference op signatures ------------------------
def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels):
    """voxel_layer.hard_voxelize (TF/mmdet3d/ops/voxel/src/voxelization.h:51-69), called as
    TF/mmdet3d/ops/voxel/voxelize.py:46-57 does (caller pre-allocates the outputs)."""
    import numpy as np
    import torch
    m = load("voxel_layer")
    pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32))
    voxels = pts.new_zeros((max_voxels, max_points, pts.shape[1]))
    coors = pts.new_zeros((max_voxels, 3), dtype=torch.int)
    num = pts.new_zeros((max_voxels,), dtype=torch.int)
    n = m.hard_voxelize(pts, voxels, coors, num, [float(v) for v in voxel_size],
                        [float(v) for v in coors_range], int(max_points), int(max_voxels), 3)
    return voxels[:n].numpy(), coors[:n].numpy(), num[:n].numpy()


def get_indice_pairs(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, subm):
    """sparse_conv_ext.get_indice_pairs_3d as called by TF/mmdet3d/ops/spconv/ops.py:46-94."""
    import numpy as np
    import torch
    from . import oracle as _o  # only for get_conv_output_size (pure arithmetic)
    m = load("sparse_conv_ext")
    ind = torch.from_numpy(np.ascontiguousarray(indices, dtype=np.int32))
    out_shape = list(spatial_shape) if subm else _o.get_conv_output_size(spatial_shape, ksize, stride, padding, dilation)
    outids, pairs, num = m.get_indice_pairs_3d(ind, int(batch_size), [int(v) for v in out_shape],
                                               [int(v) for v in spatial_shape], [int(v) for v in ksize],
                                               [int(v) for v in stride], [int(v) for v in padding],
                                               [int(v) for v in dilation], [0, 0, 0], int(bool(subm)), 0)
    return outids.numpy(), pairs.numpy(), num.numpy(), out_shape


def get_indice_pairs_transpose(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, output_padding=(0, 0, 0)):
    """sparse_conv_ext.get_indice_pairs_3d with transpose = 1 (ops.py:72-94: deconv output size, out_padding)."""
    import numpy as np
    import torch
    from . import oracle as _o  # only for get_deconv_output_size (pure arithmetic)
    m = load("sparse_conv_ext")
    ind = torch.from_numpy(np.ascontiguousarray(indices, dtype=np.int32))
    out_shape = _o.get_deconv_output_size(spatial_shape, ksize, stride, padding, dilation, list(output_padding))
    outids, pairs, num = m.get_indice_pairs_3d(ind, int(batch_size), [int(v) for v in out_shape],
                                               [int(v) for v in spatial_shape], [int(v) for v in ksize],
                                               [int(v) for v in stride], [int(v) for v in padding],
                                               [int(v) for v in dilation], [int(v) for v in output_padding], 0, 1)
    return outids.numpy(), pairs.numpy(), num.numpy(), out_shape


def indice_conv(features, filters, pairs, num, num_act_out, subm, inverse=False):
    """sparse_conv_ext.indice_conv_fp32 (TF/.../spconv_ops.h:260-361)."""
    import numpy as np
    import torch
    m = load("sparse_conv_ext")
    t = lambda a, d: torch.from_numpy(np.ascontiguousarray(a, dtype=d))
    return m.indice_conv_fp32(t(features, np.float32), t(filters, np.float32), t(pairs, np.int32),
                              t(num, np.int32), int(num_act_out), int(bool(inverse)), int(bool(subm))).numpy()


def indice_maxpool(features, pairs, num, num_act_out):
    """sparse_conv_ext.indice_maxpool_fp32 (TF/.../pool_ops.h:26-58)."""
    import numpy as np
    import torch
    m = load("sparse_conv_ext")
    t = lambda a, d: torch.from_numpy(np.ascontiguousarray(a, dtype=d))
    return m.indice_maxpool_fp32(t(features, np.float32), t(pairs, np.int32), t(num, np.int32), int(num_act_out)).numpy()


def indice_maxpool_backward(features, out_features, out_grad, pairs, num):
    """sparse_conv_ext.indice_maxpool_backward_fp32 (TF/.../pool_ops.h:60-94)."""
    import numpy as np
    import torch
    m = load("sparse_conv_ext")
    t = lambda a, d: torch.from_numpy(np.ascontiguousarray(a, dtype=d))
    return m.indice_maxpool_backward_fp32(t(features, np.float32), t(out_features, np.float32), t(out_grad, np.float32),
                                          t(pairs, np.int32), t(num, np.int32)).numpy()


def dynamic_voxelize(points, voxel_size, coors_range):
    """voxel_layer.dynamic_voxelize (TF/mmdet3d/ops/voxel/src/voxelization.h:71-86), called as voxelize.py:41-44 does."""
    import numpy as np
    import torch
    m = load("voxel_layer")
    pts = torch.from_numpy(np.ascontiguousarray(points, dtype=np.float32))
    coors = pts.new_zeros(size=(pts.size(0), 3), dtype=torch.int)
    m.dynamic_voxelize(pts, coors, [float(v) for v in voxel_size], [float(v) for v in coors_range], 3)
    return coors.numpy()


def indice_conv_backward(features, filters, out_grad, pairs, num, subm):
    """sparse_conv_ext.indice_conv_backward_fp32 (TF/.../spconv_ops.h:363-456) -> (input_grad, filters_grad)."""
    import numpy as np
    import torch
    m = load("sparse_conv_ext")
    t = lambda a, d: torch.from_numpy(np.ascontiguousarray(a, dtype=d))
    gi, gw = m.indice_conv_backward_fp32(t(features, np.float32), t(filters, np.float32), t(out_grad, np.float32),
                                         t(pairs, np.int32), t(num, np.int32), 0, int(bool(subm)))
    return gi.numpy(), gw.numpy()


def boxes_iou_bev_cpu(boxes_a, boxes_b):
    """iou3d_nms_cuda.boxes_iou_bev_cpu (CP/det3d/ops/iou3d_nms/src/iou3d_cpu.cpp:224-252): the reference's CPU path."""
    import numpy as np
    import torch
    m = load("iou3d_nms_cuda")
    a = torch.from_numpy(np.ascontiguousarray(boxes_a, dtype=np.float32))
    b = torch.from_numpy(np.ascontiguousarray(boxes_b, dtype=np.float32))
    out = torch.zeros(a.shape[0], b.shape[0])
    m.boxes_iou_bev_cpu(a, b, out)
    return out.numpy()
