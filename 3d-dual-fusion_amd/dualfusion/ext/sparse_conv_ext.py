"""`sparse_conv_ext` (TF/mmdet3d/ops/spconv/src/all.cc:21-51): rulebook construction, gather-GEMM-scatter convolution
(+ backward, + fused bias) and sparse max pooling with the reference's signatures (spconv_ops.h:27-141,143-258,260-456,
fused_spconv_ops.h:28-132, pool_ops.h:26-94), as its Python layer calls them (TF/mmdet3d/ops/spconv/ops.py:46-184).
The `*_half` entries ride the fp32 kernels (exact widening, fp32 accumulation, one rounding of the result)."""
import torch

from .. import ops as _ops
from ..spconv import ops as _sp
from ._common import need_cuda_contiguous, runtime_errors


def _pairs(ndim, indices, batch_size, out_shape, spatial_shape, ksize, stride, padding, dilation, out_padding, subm, transpose):
    need_cuda_contiguous(indices, "indices")
    if indices.dtype != torch.int32:
        raise RuntimeError("indices must be int32")
    if indices.dim() != 2 or indices.shape[1] - 1 != ndim:
        raise RuntimeError("error: indices must be [N, %d] (batch index + %d coordinates)" % (ndim + 1, ndim))
    for v in (out_shape, spatial_shape, ksize, stride, padding, dilation, out_padding):
        if len(v) != ndim:
            raise RuntimeError("error: every geometry argument needs %d entries" % ndim)
    if ndim == 2:                       # a 2-D rulebook is the 3-D one of a single-slice volume
        ind3 = torch.cat([indices[:, :1], torch.zeros_like(indices[:, :1]), indices[:, 1:]], 1).contiguous()
        out = _pairs(3, ind3, batch_size, [1] + list(out_shape), [1] + list(spatial_shape), [1] + list(ksize), [1] + list(stride),
                     [0] + list(padding), [1] + list(dilation), [0] + list(out_padding), subm, transpose)
        return [out[0][:, [0, 2, 3]].contiguous(), out[1], out[2]]
    outids, nbr, shape, _ = _sp.build_rulebook(indices, int(batch_size), [int(s) for s in spatial_shape], [int(k) for k in ksize],
                                               [int(s) for s in stride], [int(p) for p in padding], [int(d) for d in dilation],
                                               bool(subm), transpose=bool(transpose), out_padding=[int(o) for o in out_padding])
    if [int(s) for s in shape] != [int(s) for s in out_shape]:
        raise RuntimeError("outSpatialShape %s does not match the geometry (%s)" % (list(out_shape), list(shape)))
    pairs, num = _ops.nbr_to_pairs(nbr, indices.shape[0])
    return [outids, pairs, num]


@runtime_errors
def get_indice_pairs_2d(indices, batch_size, out_shape, spatial_shape, ksize, stride, padding, dilation, out_padding, subm, transpose):
    return _pairs(2, indices, batch_size, out_shape, spatial_shape, ksize, stride, padding, dilation, out_padding, subm, transpose)


@runtime_errors
def get_indice_pairs_3d(indices, batch_size, out_shape, spatial_shape, ksize, stride, padding, dilation, out_padding, subm, transpose):
    """-> [outids int32 [n_out, 4], indice_pairs int32 [K, 2, N], indice_num int32 [K]]."""
    return _pairs(3, indices, batch_size, out_shape, spatial_shape, ksize, stride, padding, dilation, out_padding, subm, transpose)


def get_indice_pairs_4d(*args, **kwargs):
    raise RuntimeError("get_indice_pairs_4d: 4-D rulebooks are not on the 3D-Dual-Fusion path (no config builds one)")


@runtime_errors
def get_indice_pairs_grid_2d(indices, grid_out, *rest):
    return get_indice_pairs_2d(indices, *rest)        # the pre-allocated dense grid is not needed (bit directory instead)


@runtime_errors
def get_indice_pairs_grid_3d(indices, grid_out, *rest):
    return get_indice_pairs_3d(indices, *rest)


def _conv_args(features, filters, indice_pairs, indice_num):
    for t, n in ((features, "features"), (filters, "filters"), (indice_pairs, "indicePairs")):
        need_cuda_contiguous(t, n)
    if indice_pairs.dtype != torch.int32 or indice_num.dtype != torch.int32:
        raise RuntimeError("indicePairs / indiceNum must be int32")


@runtime_errors
def indice_conv_fp32(features, filters, indice_pairs, indice_num, num_act_out, inverse, subm):
    _conv_args(features, filters, indice_pairs, indice_num)
    return _sp.indice_conv(features, filters, indice_pairs, indice_num, int(num_act_out), bool(inverse), bool(subm))


@runtime_errors
def indice_conv_backward_fp32(features, filters, out_grad, indice_pairs, indice_num, inverse, subm):
    _conv_args(features, filters, indice_pairs, indice_num)
    need_cuda_contiguous(out_grad, "outGrad")
    return _sp.indice_conv_backward(features, filters, out_grad, indice_pairs, indice_num, bool(inverse), bool(subm))


@runtime_errors
def fused_indice_conv_fp32(features, filters, bias, indice_pairs, indice_num, num_act_out, inverse, subm):
    _conv_args(features, filters, indice_pairs, indice_num)
    need_cuda_contiguous(bias, "bias")
    return _sp.fused_indice_conv(features, filters, bias, indice_pairs, indice_num, int(num_act_out), bool(inverse), bool(subm))


@runtime_errors
def indice_maxpool_fp32(features, indice_pairs, indice_num, num_act):
    need_cuda_contiguous(features, "features")
    return _sp.indice_maxpool(features, indice_pairs.contiguous(), indice_num, int(num_act))


@runtime_errors
def indice_maxpool_backward_fp32(features, out_features, out_grad, indice_pairs, indice_num):
    for t, n in ((features, "features"), (out_features, "outFeatures"), (out_grad, "outGrad")):
        need_cuda_contiguous(t, n)
    return _sp.indice_maxpool_backward(features, out_features, out_grad, indice_pairs.contiguous(), indice_num)


indice_conv_half, indice_conv_backward_half = indice_conv_fp32, indice_conv_backward_fp32
fused_indice_conv_half = fused_indice_conv_fp32
indice_maxpool_half, indice_maxpool_backward_half = indice_maxpool_fp32, indice_maxpool_backward_fp32
