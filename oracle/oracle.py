"""oracle/oracle.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (plain C via ctypes for the integer work, numpy for the fp32
arithmetic) of the reference algorithms on the 3D-Dual-Fusion hot path.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the product package never does and fails loudly without its HIP library.

Pinned (tests/test_oracle_*.py) against
  * the reference's own compiled CPU code in oracle/_ref (oracle/build_ref.py) and
  * the golden vectors committed under tests/golden/ (generated from the reference
    by tests/golden/make_golden.py).

Path shorthand in citations: TF/ = /root/reference/TransFusion, CP/ = /root/reference/CenterPoint.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """gcc-compile oracle/df3d_oracle.c -> oracle/libdf3d_oracle.so."""
    src = os.path.join(_HERE, "df3d_oracle.c")
    out = os.path.join(_HERE, "libdf3d_oracle.so")
    if force or not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-ffp-contract=off", "-o", out, src, "-lm"])
    return out


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


_f32 = ctypes.c_float
_i32 = ctypes.c_int32


# --------------------------------------------------------------------- voxelize
def hard_voxelize(points, voxel_size, coors_range, max_points, max_voxels, variant="cpp"):
    """TF/mmdet3d/ops/voxel/src/voxelization_cpu.cpp:43-141 (variant='cpp') or
    CP/det3d/ops/point_cloud/point_cloud_ops.py:7-55 (variant='numba').
    Returns voxels [M,max_points,C] f32, coors [M,3] i32 (z,y,x), num [M] i32."""
    points = np.ascontiguousarray(points, dtype=np.float32)
    P, C = points.shape
    cap = max_voxels if max_voxels != -1 else P
    voxels = np.zeros((cap, max_points, C), np.float32)
    coors = np.zeros((cap, 3), np.int32)
    num = np.zeros((cap,), np.int32)
    vs = np.asarray(voxel_size, np.float32)
    rg = np.asarray(coors_range, np.float32)
    fn = lib().orc_hard_voxelize if variant == "cpp" else lib().orc_points_to_voxel_numba
    n = fn(_p(points, _f32), P, C, _p(vs, _f32), _p(rg, _f32), int(max_points), int(max_voxels),
           _p(voxels, _f32), _p(coors, _i32), _p(num, _i32))
    assert n >= 0
    return voxels[:n], coors[:n], num[:n]


def mean_vfe(voxels, num, clamp_min=None):
    """CP/det3d/models/readers/voxel_encoder.py:17-24 (sum over the padded point axis /
    num_points); VR/pcdet/models/backbones_3d/vfe/mean_vfe.py:24-29 clamps num >= 1."""
    n = num.astype(np.float32)
    if clamp_min is not None:
        n = np.maximum(n, np.float32(clamp_min))
    return (voxels.sum(axis=1, dtype=np.float32) / n[:, None]).astype(np.float32)


# --------------------------------------------------------------------- rulebook
def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    """TF/mmdet3d/ops/spconv/ops.py:20-30."""
    return [(input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) // stride[i] + 1
            for i in range(len(input_size))]


def get_deconv_output_size(input_size, kernel_size, stride, padding, dilation, output_padding):
    """TF/mmdet3d/ops/spconv/ops.py:33-44."""
    return [(input_size[i] - 1) * stride[i] - 2 * padding[i] + kernel_size[i] + output_padding[i]
            for i in range(len(input_size))]


def get_indice_pairs_transpose(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, output_padding=(0, 0, 0)):
    """get_indice_pairs(..., transpose=True) (ops.py:46-94 -> spconv_ops.h:27-141 -> geometry.h:88-142,194-245, CPU path).
    Returns (outids, indice_pairs [K,2,N], indice_num [K], out_shape) in the reference CPU order."""
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    N = indices.shape[0]
    K = int(np.prod(ksize))
    out_shape = get_deconv_output_size(spatial_shape, ksize, stride, padding, dilation, list(output_padding))
    outids = np.zeros((max(N * K, 1), 4), np.int32)
    pairs = np.empty((K, 2, max(N, 1)), np.int32)
    num = np.zeros((K,), np.int32)
    arr = lambda v: np.asarray(v, np.int32)
    os_, ks_, st_, pd_, dl_ = arr(out_shape), arr(ksize), arr(stride), arr(padding), arr(dilation)
    n_out = lib().orc_get_indice_pairs_transpose(_p(indices, _i32), N, int(batch_size), _p(os_, _i32), _p(ks_, _i32),
                                                 _p(st_, _i32), _p(pd_, _i32), _p(dl_, _i32), _p(outids, _i32),
                                                 _p(pairs, _i32), _p(num, _i32))
    assert n_out >= 0
    return outids[:n_out].copy(), pairs[:, :, :N], num, out_shape


def get_indice_pairs(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, subm):
    """TF/mmdet3d/ops/spconv/ops.py:46-94 -> spconv_ops.h:27-141 -> geometry.h:144-297 (CPU path).
    Returns (outids [N_out,4], indice_pairs [K,2,N] (-1 padded), indice_num [K], out_shape),
    in the reference CPU order (out voxels by first touch, pairs in input order)."""
    indices = np.ascontiguousarray(indices, dtype=np.int32)
    N = indices.shape[0]
    K = int(np.prod(ksize))
    if subm:
        out_shape = list(spatial_shape)
    else:
        out_shape = get_conv_output_size(spatial_shape, ksize, stride, padding, dilation)
    outids = np.zeros((max(N * K, 1), 4), np.int32)
    pairs = np.empty((K, 2, max(N, 1)), np.int32)
    num = np.zeros((K,), np.int32)
    arr = lambda v: np.asarray(v, np.int32)
    os_, ks_, st_, pd_, dl_ = arr(out_shape), arr(ksize), arr(stride), arr(padding), arr(dilation)
    n_out = lib().orc_get_indice_pairs(_p(indices, _i32), N, int(batch_size), _p(os_, _i32), _p(ks_, _i32),
                                       _p(st_, _i32), _p(pd_, _i32), _p(dl_, _i32), int(bool(subm)),
                                       _p(outids, _i32), _p(pairs, _i32), _p(num, _i32))
    assert n_out >= 0
    pairs = pairs[:, :, :N]
    if subm:
        return indices, pairs, num, out_shape
    return outids[:n_out].copy(), pairs, num, out_shape


def canonical_rulebook(outids, pairs, num):
    """Order-independent form of a rulebook (SURVEY.md §8c 'canonical order'): out voxels sorted
    by (b,z,y,x); per offset the (in, relabelled-out) pairs sorted.  Returns
    (outids_sorted, [K arrays of shape [n_k,2]])."""
    outids = np.asarray(outids)
    order = np.lexsort((outids[:, 3], outids[:, 2], outids[:, 1], outids[:, 0]))
    relabel = np.empty(len(order), np.int64)
    relabel[order] = np.arange(len(order))
    lists = []
    for k in range(pairs.shape[0]):
        n = int(num[k])
        i = pairs[k, 0, :n].astype(np.int64)
        o = relabel[pairs[k, 1, :n].astype(np.int64)] if n else np.zeros(0, np.int64)
        pr = np.stack([i, o], 1)
        pr = pr[np.lexsort((pr[:, 1], pr[:, 0]))]
        lists.append(pr)
    return outids[order], lists


# --------------------------------------------------------------------- sparse conv
def indice_conv(features, filters, pairs, num, num_act_out, subm, inverse=False):
    """TF/mmdet3d/ops/spconv/include/spconv/spconv_ops.h:260-361 (indiceConv<float>):
    output = zeros; subM: output = features @ W[argmax num] first (:300-303); then for every
    other non-empty offset: gather (src/reordering.cc:20-33) -> mm -> scatter-add (:35-50)."""
    features = np.asarray(features, np.float32)
    K = pairs.shape[0]
    cin, cout = filters.shape[-2], filters.shape[-1]
    W = np.asarray(filters, np.float32).reshape(K, cin, cout)
    out = np.zeros((num_act_out, cout), np.float32)
    kmax = int(np.argmax(num))
    if subm:
        out[:] = features @ W[kmax]
    a, b = (1, 0) if inverse else (0, 1)
    for k in range(K):
        n = int(num[k])
        if n <= 0 or (subm and k == kmax):
            continue
        buf = features[pairs[k, a, :n]] @ W[k]
        out[pairs[k, b, :n]] += buf  # out rows are unique within one offset
    return out


def indice_conv_backward(features, filters, out_grad, pairs, num, subm, inverse=False):
    """TF/mmdet3d/ops/spconv/include/spconv/spconv_ops.h:363-456 (indiceConvBackward<float>): per non-empty offset
    filtersGrad[k] = gather(features)^T @ gather(outGrad), inputGrad[in rows] += gather(outGrad) @ W[k]^T; subM handles
    the offset with the most pairs (the centre) with two dense GEMMs first (:398-402).  float64 accumulation.
    -> (input_grad [N_in, Cin], filters_grad (shape of filters))."""
    features = np.asarray(features, np.float64)
    og = np.asarray(out_grad, np.float64)
    K = pairs.shape[0]
    cin, cout = filters.shape[-2], filters.shape[-1]
    W = np.asarray(filters, np.float64).reshape(K, cin, cout)
    gin = np.zeros_like(features)
    gw = np.zeros_like(W)
    kmax = int(np.argmax(num))
    if subm:
        gw[kmax] = features.T @ og
        gin[:] = og @ W[kmax].T
    a, b = (1, 0) if inverse else (0, 1)
    for k in range(K):
        n = int(num[k])
        if n <= 0 or (subm and k == kmax):
            continue
        ib, ob = features[pairs[k, a, :n]], og[pairs[k, b, :n]]
        gw[k] = ib.T @ ob
        np.add.at(gin, pairs[k, a, :n], ob @ W[k].T)
    return gin.astype(np.float32), gw.reshape(np.shape(filters)).astype(np.float32)


def batchnorm_eval(x, weight, bias, mean, var, eps):
    """nn.BatchNorm1d in eval mode (eps 1e-3 everywhere on the path, CP/.../scn.py:108-109)."""
    return ((x - mean) / np.sqrt(var + np.float32(eps)) * weight + bias).astype(np.float32)


def dense(features, indices, spatial_shape, batch_size):
    """TF/mmdet3d/ops/spconv/structure.py:5-18,55-64: zeros [B,*S,C] <- index_put, permute to [B,C,*S]."""
    C = features.shape[1]
    res = np.zeros([batch_size] + list(spatial_shape) + [C], np.float32)
    ind = np.asarray(indices, np.int64)
    res[ind[:, 0], ind[:, 1], ind[:, 2], ind[:, 3]] = features
    return np.ascontiguousarray(res.transpose(0, 4, 1, 2, 3))


# --------------------------------------------------------------------- MSDA
def ms_deform_attn(value, spatial_shapes, sampling_locations, attention_weights):
    """Multi-scale deformable attention forward.  Restates the semantics of
    CP/det3d/models/model_utils/ops/functions/ms_deform_attn_func.py:41-61
    (grid_sample bilinear, align_corners=False, zero padding) == the CUDA kernel
    ops/src/cuda/ms_deform_im2col_cuda.cuh:237-299 (h_im = loc_h*H - 0.5, corners outside
    contribute 0).  value [N,S,M,D]; shapes [(H,W)..]; loc [N,Lq,M,L,P,2] (x,y in [0,1]);
    weights [N,Lq,M,L,P] -> out [N,Lq,M*D]."""
    value = np.asarray(value, np.float32)
    loc = np.asarray(sampling_locations, np.float32)
    aw = np.asarray(attention_weights, np.float32)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    out = np.zeros((N, Lq, M, D), np.float32)
    start = 0
    nn = np.arange(N)[:, None, None, None]
    mm = np.arange(M)[None, None, :, None]
    for l, (H, W) in enumerate(spatial_shapes):
        H, W = int(H), int(W)
        v = value[:, start:start + H * W].reshape(N, H, W, M, D)
        start += H * W
        w_im = loc[:, :, :, l, :, 0] * np.float32(W) - np.float32(0.5)  # [N,Lq,M,P]
        h_im = loc[:, :, :, l, :, 1] * np.float32(H) - np.float32(0.5)
        h0 = np.floor(h_im).astype(np.int64)
        w0 = np.floor(w_im).astype(np.int64)
        lh = (h_im - h0).astype(np.float32)
        lw = (w_im - w0).astype(np.float32)
        acc = np.zeros((N, Lq, M, P, D), np.float32)
        for dh, dw, wt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw),
                           (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
            hh, ww = h0 + dh, w0 + dw
            ok = (hh >= 0) & (hh < H) & (ww >= 0) & (ww < W)
            hc, wc = np.clip(hh, 0, H - 1), np.clip(ww, 0, W - 1)
            g = v[nn, hc, wc, mm]  # [N,Lq,M,P,D]
            acc += g * (wt * ok)[..., None].astype(np.float32)
        out += (acc * aw[:, :, :, l, :, None]).sum(axis=3, dtype=np.float32)
    return out.reshape(N, Lq, M * D)


def ms_deform_attn_backward(value, spatial_shapes, sampling_locations, attention_weights, grad_output):
    """Backward of ms_deform_attn.  Restates ms_deformable_col2im_gpu_kernel* + ms_deform_attn_col2im_bilinear
    (CP/det3d/models/model_utils/ops/src/cuda/ms_deform_im2col_cuda.cuh:87-232,301-921) in float64 accumulation:
    grad_value[corner] += w_corner * attn * g; grad_attn = sum_c g * bilinear; grad_loc = (W * dval/dw, H * dval/dh)
    * attn * g summed over channels; samples with h_im / w_im outside (-1, H) / (-1, W) contribute nothing.
    -> (grad_value [N,S,M,D], grad_loc [N,Lq,M,L,P,2], grad_attn [N,Lq,M,L,P]) float32."""
    value = np.asarray(value, np.float64)
    loc = np.asarray(sampling_locations, np.float32)
    aw = np.asarray(attention_weights, np.float64)
    N, S, M, D = value.shape
    _, Lq, _, L, P, _ = loc.shape
    g = np.asarray(grad_output, np.float64).reshape(N, Lq, M, 1, D)
    gv = np.zeros((N, S, M, D), np.float64)
    gl = np.zeros((N, Lq, M, L, P, 2), np.float64)
    ga = np.zeros((N, Lq, M, L, P), np.float64)
    nn = np.broadcast_to(np.arange(N)[:, None, None, None], (N, Lq, M, P))
    mm = np.broadcast_to(np.arange(M)[None, None, :, None], (N, Lq, M, P))
    start = 0
    for l, (H, W) in enumerate(spatial_shapes):
        H, W = int(H), int(W)
        v = value[:, start:start + H * W].reshape(N, H, W, M, D)
        gvl = np.zeros((N, H, W, M, D), np.float64)
        w_im = loc[:, :, :, l, :, 0] * np.float32(W) - np.float32(0.5)
        h_im = loc[:, :, :, l, :, 1] * np.float32(H) - np.float32(0.5)
        inside = (h_im > -1) & (w_im > -1) & (h_im < H) & (w_im < W)
        h0 = np.floor(h_im).astype(np.int64)
        w0 = np.floor(w_im).astype(np.int64)
        lh = (h_im - h0).astype(np.float32).astype(np.float64)
        lw = (w_im - w0).astype(np.float32).astype(np.float64)
        hh, hw = 1 - lh, 1 - lw
        corner = []
        for dh, dw, wt in ((0, 0, hh * hw), (0, 1, hh * lw), (1, 0, lh * hw), (1, 1, lh * lw)):
            y, x = h0 + dh, w0 + dw
            ok = inside & (y >= 0) & (y < H) & (x >= 0) & (x < W)
            yc, xc = np.clip(y, 0, H - 1), np.clip(x, 0, W - 1)
            vals = v[nn, yc, xc, mm] * ok[..., None]                                 # [N,Lq,M,P,D]
            corner.append(vals)
            contrib = (wt * ok * aw[:, :, :, l, :])[..., None] * g                   # [N,Lq,M,P,D]
            np.add.at(gvl, (nn, yc, xc, mm), contrib)
        v1, v2, v3, v4 = corner
        val = (hh * hw)[..., None] * v1 + (hh * lw)[..., None] * v2 + (lh * hw)[..., None] * v3 + (lh * lw)[..., None] * v4
        dh_ = hw[..., None] * (v3 - v1) + lw[..., None] * (v4 - v2)
        dw_ = hh[..., None] * (v2 - v1) + lh[..., None] * (v4 - v3)
        tg = g * aw[:, :, :, l, :, None]
        ga[:, :, :, l, :] = (g * val).sum(-1) * inside
        gl[:, :, :, l, :, 0] = W * (dw_ * tg).sum(-1) * inside
        gl[:, :, :, l, :, 1] = H * (dh_ * tg).sum(-1) * inside
        gv[:, start:start + H * W] = gvl.reshape(N, H * W, M, D)
        start += H * W
    return gv.astype(np.float32), gl.astype(np.float32), ga.astype(np.float32)


# --------------------------------------------------------------------- point ops
def furthest_point_sample(xyz, m):
    """CP/det3d/ops/furthest_point_sample/src/furthest_point_sample_cuda.cu:26-141."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    B, N, _ = xyz.shape
    idx = np.zeros((B, m), np.int32)
    lib().orc_fps(_p(xyz, _f32), B, N, int(m), _p(idx, _i32))
    return idx


def ball_query(min_radius, max_radius, nsample, xyz, new_xyz):
    """CP/det3d/ops/ball_query/src/ball_query_cuda.cu:11-54 (idx zero-initialised by the
    wrapper ball_query.py)."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    new_xyz = np.ascontiguousarray(new_xyz, np.float32)
    B, N, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = np.zeros((B, m, nsample), np.int32)
    lib().orc_ball_query.argtypes = [ctypes.POINTER(_f32), ctypes.POINTER(_f32), ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, _f32, _f32, ctypes.c_int, ctypes.POINTER(_i32)]
    lib().orc_ball_query(_p(new_xyz, _f32), _p(xyz, _f32), B, N, m, float(min_radius), float(max_radius),
                         int(nsample), _p(idx, _i32))
    return idx


def group_points(features, idx):
    """CP/det3d/ops/group_points/src/group_points_cuda.cu:56-78: out[b,c,p,s] = feat[b,c,idx[b,p,s]]."""
    B = features.shape[0]
    return np.stack([features[b][:, idx[b]] for b in range(B)], 0)


def gather_points(features, idx):
    """CP/det3d/ops/gather_points/src/gather_points_cuda.cu:8-24: out[b,c,p] = feat[b,c,idx[b,p]]."""
    B = features.shape[0]
    return np.stack([features[b][:, idx[b]] for b in range(B)], 0)


# --------------------------------------------------------------------- rotated BEV IoU / NMS (detection tail)
def boxes_pairwise_bev(boxes_a, boxes_b, mode="iou"):
    """CP/det3d/ops/iou3d_nms/src/iou3d_cpu.cpp:125-252: [N,7] x [M,7] -> overlap area ('overlap') or rotated IoU."""
    a = np.ascontiguousarray(boxes_a, np.float32)
    b = np.ascontiguousarray(boxes_b, np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    lib().orc_boxes_pairwise(_p(a, _f32), a.shape[0], _p(b, _f32), b.shape[0], 1 if mode == "iou" else 0, _p(out, _f32))
    return out


def tf_boxes_overlap_bev(boxes_a, boxes_b):
    """TF/mmdet3d/ops/iou3d/src/iou3d_kernel.cu:122-240 + iou3d.cpp:66-90 (boxes_overlap_bev_gpu): [N,5] x [M,5]
    boxes (x1, y1, x2, y2, angle) -> overlap areas [N,M]."""
    a = np.ascontiguousarray(boxes_a, np.float32)
    b = np.ascontiguousarray(boxes_b, np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    if a.shape[0] and b.shape[0]:
        lib().orc_tf_boxes_overlap_bev(_p(a, _f32), a.shape[0], _p(b, _f32), b.shape[0], _p(out, _f32))
    return out


def tf_bbox_overlaps_3d(boxes1, boxes2):
    """BboxOverlaps3D(coordinate='lidar') of the TransFusion tree: TF/mmdet3d/core/bbox/iou_calculators/
    iou3d_calculator.py:138-166 -> LiDARInstance3DBoxes.overlaps (core/bbox/structures/base_box3d.py:352-438):
    boxes (x, y, z_bottom, w, l, h, yaw, ...) -> 3-D IoU [N,M] in fp32."""
    f32 = np.float32
    b1, b2 = np.asarray(boxes1, f32)[:, :7], np.asarray(boxes2, f32)[:, :7]
    if len(b1) * len(b2) == 0:
        return np.zeros((len(b1), len(b2)), f32)

    def xyxyr(b):                                            # bev (lidar_box3d.py:87-90) -> xywhr2xyxyr (utils.py:64-82)
        hw, hl = b[:, 3] / f32(2), b[:, 4] / f32(2)
        return np.stack([b[:, 0] - hw, b[:, 1] - hl, b[:, 0] + hw, b[:, 1] + hl, b[:, 6]], 1).astype(f32)
    top1, bot1 = (b1[:, 2] + b1[:, 5])[:, None], b1[:, 2][:, None]
    top2, bot2 = (b2[:, 2] + b2[:, 5])[None], b2[:, 2][None]
    oh = np.maximum(np.minimum(top1, top2) - np.maximum(bot1, bot2), f32(0))
    o3 = (tf_boxes_overlap_bev(xyxyr(b1), xyxyr(b2)) * oh).astype(f32)
    v1 = (b1[:, 3] * b1[:, 4] * b1[:, 5])[:, None]
    v2 = (b2[:, 3] * b2[:, 4] * b2[:, 5])[None]
    return (o3 / np.maximum(v1 + v2 - o3, f32(1e-8))).astype(f32)


def nms_bev(boxes_sorted, thresh, rotated=True, margin=0.0):
    """iou3d_nms.cpp:88-139 (nms_gpu) / :142-188 (nms_normal_gpu) on boxes already sorted by descending score.
    Returns (keep indices int64, number of evaluated IoUs within `margin` of the threshold)."""
    b = np.ascontiguousarray(boxes_sorted, np.float32)
    keep = np.zeros((max(b.shape[0], 1),), np.int64)
    close = ctypes.c_int(0)
    f = lib().orc_nms_bev
    f.argtypes = [ctypes.POINTER(_f32), ctypes.c_int, _f32, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), _f32,
                  ctypes.POINTER(ctypes.c_int)]
    f.restype = ctypes.c_int
    mode = 2 if rotated == "circle" else int(bool(rotated))
    n = f(_p(b, _f32), b.shape[0], float(thresh), mode, _p(keep, ctypes.c_int64), float(margin),
          ctypes.byref(close))
    return keep[:n].copy(), int(close.value)


def rotate_nms_pcdet(boxes, scores, thresh, pre_maxsize=None, post_max_size=None, margin=0.0):
    """CP/det3d/core/bbox/box_torch_ops.py:248-279: column swap + heading flip into pcdet's frame, stable sort by
    descending score (ties: lower index first -- the reference's torch sort leaves ties unspecified), rotated NMS."""
    boxes = np.asarray(boxes, np.float32)[:, [0, 1, 2, 4, 3, 5, -1]].copy()
    boxes[:, -1] = -boxes[:, -1] - np.float32(np.pi / 2)
    order = np.argsort(-np.asarray(scores, np.float32), kind="stable")
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    if len(order) == 0:
        return order.astype(np.int64), 0
    keep, close = nms_bev(boxes[order], thresh, True, margin)
    sel = order[keep]
    if post_max_size is not None:
        sel = sel[:post_max_size]
    return sel.astype(np.int64), close


def centerhead_predict(preds_dicts, test_cfg, num_classes, margin_out=None):
    """CenterHead.predict + post_processing (CP/det3d/models/bbox_heads/center_head.py:302-501, no double flip,
    rotate NMS or circular NMS) restated with numpy float32 arithmetic in the reference's evaluation order.
    preds_dicts: per task {'hm','reg','height','dim','rot'[,'vel']: [B, C, H, W] float32}.
    test_cfg keys: post_center_limit_range, score_threshold, pc_range, out_size_factor, voxel_size,
    nms{nms_pre_max_size, nms_post_max_size, nms_iou_threshold} [, circular_nms, min_radius].
    Returns per sample {'box3d_lidar','scores','label_preds'}; margin_out (dict) collects how close the decisive
    comparisons came to flipping ('score_gap', 'thr_gap', 'range_gap', 'iou_close')."""
    f32 = np.float32
    rng = np.asarray(test_cfg["post_center_limit_range"], f32)
    stats = dict(score_gap=np.inf, thr_gap=np.inf, range_gap=np.inf, iou_close=0)
    rets = []
    for task_id, pd in enumerate(preds_dicts):
        pd = {k: np.ascontiguousarray(np.transpose(np.asarray(v, f32), (0, 2, 3, 1))) for k, v in pd.items()}
        B, H, W, ncls = pd["hm"].shape
        hm = (f32(1) / (f32(1) + np.exp(-pd["hm"]))).astype(f32)
        dim = np.exp(pd["dim"]).astype(f32)
        rot = np.arctan2(pd["rot"][..., 0:1], pd["rot"][..., 1:2]).astype(f32)
        ys, xs = np.meshgrid(np.arange(H, dtype=f32), np.arange(W, dtype=f32), indexing="ij")
        xs = xs.reshape(1, -1, 1) + pd["reg"].reshape(B, H * W, 2)[:, :, 0:1]
        ys = ys.reshape(1, -1, 1) + pd["reg"].reshape(B, H * W, 2)[:, :, 1:2]
        xs = xs * f32(test_cfg["out_size_factor"]) * f32(test_cfg["voxel_size"][0]) + f32(test_cfg["pc_range"][0])
        ys = ys * f32(test_cfg["out_size_factor"]) * f32(test_cfg["voxel_size"][1]) + f32(test_cfg["pc_range"][1])
        parts = [xs, ys, pd["height"].reshape(B, H * W, 1), dim.reshape(B, H * W, 3)]
        if "vel" in pd:
            parts.append(pd["vel"].reshape(B, H * W, 2))
        parts.append(rot.reshape(B, H * W, 1))
        boxes_all = np.concatenate(parts, axis=2).astype(f32)
        hm = hm.reshape(B, H * W, ncls)
        per_sample = []
        for i in range(B):
            box_preds, hm_preds = boxes_all[i], hm[i]
            labels = np.argmax(hm_preds, axis=-1)                  # first maximum, like torch.max
            scores = hm_preds[np.arange(len(labels)), labels]
            smask = scores > f32(test_cfg["score_threshold"])
            dmask = (box_preds[:, :3] >= rng[:3]).all(1) & (box_preds[:, :3] <= rng[3:]).all(1)
            stats["thr_gap"] = min(stats["thr_gap"], float(np.abs(scores - f32(test_cfg["score_threshold"])).min()))
            stats["range_gap"] = min(stats["range_gap"], float(np.minimum(np.abs(box_preds[:, :3] - rng[:3]),
                                                                         np.abs(box_preds[:, :3] - rng[3:])).min()))
            mask = smask & dmask
            box_preds, scores, labels = box_preds[mask], scores[mask], labels[mask]
            if len(scores) > 1:
                stats["score_gap"] = min(stats["score_gap"], float(np.diff(np.sort(scores)).min()))
            nms = test_cfg["nms"]
            if test_cfg.get("circular_nms", False):
                order = np.argsort(-scores, kind="stable")
                b7 = np.zeros((len(order), 7), f32)
                b7[:, :2] = box_preds[order][:, :2]
                keep, _ = nms_bev(b7, test_cfg["min_radius"][task_id], "circle")
                sel = order[keep][:nms["nms_post_max_size"]]
            else:
                sel, close = rotate_nms_pcdet(box_preds[:, [0, 1, 2, 3, 4, 5, -1]], scores, nms["nms_iou_threshold"],
                                              nms["nms_pre_max_size"], nms["nms_post_max_size"], margin=1e-4)
                stats["iou_close"] += close
            per_sample.append(dict(box3d_lidar=box_preds[sel], scores=scores[sel], label_preds=labels[sel].astype(np.int64)))
        rets.append(per_sample)
    out = []
    for i in range(len(rets[0])):
        flag, lab = 0, []
        for j, nc in enumerate(num_classes):
            lab.append(rets[j][i]["label_preds"] + flag)
            flag += nc
        out.append(dict(box3d_lidar=np.concatenate([r[i]["box3d_lidar"] for r in rets]),
                        scores=np.concatenate([r[i]["scores"] for r in rets]), label_preds=np.concatenate(lab)))
    if margin_out is not None:
        margin_out.update(stats)
    return out


def indice_maxpool(features, pairs, num, num_act_out):
    """TF/mmdet3d/ops/spconv/include/spconv/pool_ops.h:26-58 + src/maxpool.cc:22-41: output = ZEROS, then per offset and
    pair out[o] = in[i] where out[o] < in[i]."""
    features = np.asarray(features, np.float32)
    out = np.zeros((num_act_out, features.shape[1]), np.float32)
    for k in range(pairs.shape[0]):
        n = int(num[k])
        if n > 0:
            np.maximum.at(out, pairs[k, 1, :n], features[pairs[k, 0, :n]])
    return out


def indice_maxpool_backward(features, out_features, out_grad, pairs, num):
    """pool_ops.h:60-94 + src/maxpool.cc:43-66: din[i] += dout[o] for every pair with out[o] == in[i], offsets in
    ascending order (an input row occurs at most once per offset, so the order of the fp32 additions is fixed)."""
    features = np.asarray(features, np.float32)
    gin = np.zeros_like(features)
    for k in range(pairs.shape[0]):
        n = int(num[k])
        if n > 0:
            i, o = pairs[k, 0, :n], pairs[k, 1, :n]
            gin[i] = (gin[i] + np.where(out_features[o] == features[i], out_grad[o], np.float32(0))).astype(np.float32)
    return gin


def dynamic_voxelize(points, voxel_size, coors_range):
    """TF/mmdet3d/ops/voxel/src/voxelization_cpu.cpp:8-41,147-171: c = floor((p - min) / size) in fp32 per axis,
    (z, y, x) order, all -1 when any axis falls outside grid = round((max - min) / size)."""
    pts = np.asarray(points, np.float32)[:, :3]
    vs = np.asarray(voxel_size, np.float32)
    rng = np.asarray(coors_range, np.float32)
    grid = np.round((rng[3:] - rng[:3]) / vs).astype(np.int64)
    with np.errstate(invalid="ignore"):
        c = np.floor((pts - rng[:3]) / vs)
        ok = ((c >= 0) & (c < grid)).all(1)
    out = np.full((len(pts), 3), -1, np.int32)
    out[ok] = c[ok][:, ::-1].astype(np.int32)
    return out
