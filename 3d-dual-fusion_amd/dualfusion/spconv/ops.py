"""Function-level API with the reference's names and argument meaning
(TF/mmdet3d/ops/spconv/ops.py:20-126), running on the MI355X kernels.

`get_indice_pairs` returns the reference-format rulebook (outids, indice_pairs [K,2,N] int32
-1 padded, indice_pair_num [K]).  Pair order inside one offset is by output row (the
reference's GPU order is atomics-dependent, its CPU order is by input row; SURVEY.md §8c fixes
the canonical comparison).  Strided-conv outputs come sorted by flat index like the
reference's GPU path (spconv_ops.h:119-137)."""
import torch

from .. import ops as _ops
from .._lib import Df3dError


def get_conv_output_size(input_size, kernel_size, stride, padding, dilation):
    ndim = len(input_size)
    output_size = []
    for i in range(ndim):
        size = (input_size[i] + 2 * padding[i] - dilation[i] * (kernel_size[i] - 1) - 1) // stride[i] + 1
        if kernel_size[i] == -1:
            output_size.append(1)
        else:
            output_size.append(size)
    return output_size


def get_deconv_output_size(input_size, kernel_size, stride, padding, dilation, output_padding):
    ndim = len(input_size)
    output_size = []
    for i in range(ndim):
        if kernel_size[i] == -1:
            raise ValueError("deconv don't support kernel_size < 0")
        size = (input_size[i] - 1) * stride[i] - 2 * padding[i] + kernel_size[i] + output_padding[i]
        output_size.append(size)
    return output_size


def build_rulebook(indices, batch_size, spatial_shape, ksize, stride, padding, dilation, subm, directory=None,
                   transpose=False, out_padding=(0, 0, 0)):
    """-> (outids, nbr [K,n_out], out_shape, out_directory or None).  Kernel-facing form.  transpose: the rulebook of a
    transposed convolution (ops.py:72-94: deconv output size; geometry.h:88-142)."""
    indices = indices.contiguous()
    if len(spatial_shape) != 3:
        raise Df3dError("only 3-D sparse convolutions are implemented on the MI355X path")
    if directory is None:
        directory = _ops.grid_build(indices, batch_size, spatial_shape)
    if subm:
        nbr = _ops.subm_neighbors(directory, indices, ksize, dilation)
        return indices, nbr, list(spatial_shape), directory
    if transpose:
        out_shape = get_deconv_output_size(spatial_shape, ksize, stride, padding, dilation, list(out_padding))
    else:
        out_shape = get_conv_output_size(spatial_shape, ksize, stride, padding, dilation)
    outids, out_dir = _ops.conv_out_indices(indices, batch_size, spatial_shape, out_shape, ksize, stride, padding,
                                            dilation, transpose=transpose)
    nbr = _ops.conv_neighbors(directory, outids, ksize, stride, padding, dilation, transpose=transpose)
    return outids, nbr, out_shape, out_dir


def get_indice_pairs(indices, batch_size, spatial_shape, ksize=3, stride=1, padding=0, dilation=1, out_padding=0,
                     subm=False, transpose=False, grid=None):
    ndim = indices.shape[1] - 1
    lst = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * ndim
    ksize, stride, padding, dilation = lst(ksize), lst(stride), lst(padding), lst(dilation)
    out_padding = lst(out_padding)
    outids, nbr, _, _ = build_rulebook(indices.int(), batch_size, spatial_shape, ksize, stride, padding, dilation, subm,
                                       transpose=bool(transpose), out_padding=out_padding)
    pairs, num = _ops.nbr_to_pairs(nbr, indices.shape[0])
    return outids, pairs, num


def _half_entry(fn):
    """The reference registers every op twice, `*_fp32` and `*_half` (TF/mmdet3d/ops/spconv/src/all.cc:21-51, dispatched
    on the dtype in ops.py:112-184).  Here fp16 tensors ride the fp32 kernels: rows and filters are widened (exact), the
    products accumulate in fp32 (the reference's half GEMM rounds its accumulator to fp16 at least once per offset), and
    every floating-point result is rounded to fp16 once."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        half = any(isinstance(a, torch.Tensor) and a.dtype == torch.float16 for a in args)
        if not half:
            return fn(*args, **kwargs)
        args = [a.float() if isinstance(a, torch.Tensor) and a.dtype == torch.float16 else a for a in args]
        out = fn(*args, **kwargs)
        if isinstance(out, (list, tuple)):
            return type(out)(o.half() if isinstance(o, torch.Tensor) and o.is_floating_point() else o for o in out)
        return out.half()
    return wrapped


@_half_entry
def indice_conv(features, filters, indice_pairs, indice_pair_num, num_activate_out, inverse=False, subm=False):
    """sparse_conv_ext.indice_conv_fp32 (spconv_ops.h:260-361) from a reference-format rulebook."""
    if features.dtype != torch.float32:
        raise Df3dError("indice_conv: fp32 only (got %s)" % features.dtype)
    pairs = indice_pairs.contiguous()
    if inverse:
        pairs = pairs.flip(1).contiguous()
    nbr = _ops.pairs_to_nbr(pairs, indice_pair_num, int(num_activate_out))
    cin = features.shape[1]
    cout = filters.shape[-1]
    return _ops.sparse_conv_fused(features.contiguous(), filters.contiguous().view(-1, cin, cout), nbr,
                                  int(num_activate_out))


@_half_entry
def indice_conv_backward(features, filters, out_bp, indice_pairs, indice_pair_num, inverse=False, subm=False):
    """sparse_conv_ext.indice_conv_backward_fp32 (spconv_ops.h:363-456) from a reference-format rulebook:
    -> [input_grad [N_in, Cin], filters_grad (shape of `filters`)]."""
    if features.dtype != torch.float32:
        raise Df3dError("indice_conv_backward: fp32 only (got %s)" % features.dtype)
    pairs = indice_pairs.contiguous()
    if inverse:
        pairs = pairs.flip(1).contiguous()
    n_out = out_bp.shape[0]
    nbr = _ops.pairs_to_nbr(pairs, indice_pair_num, int(n_out))
    cin, cout = features.shape[1], filters.shape[-1]
    g_in, g_w = _ops.sparse_conv_backward(features.contiguous(), filters.contiguous().view(-1, cin, cout),
                                          out_bp.contiguous(), nbr, False)
    return [g_in, g_w.view_as(filters)]


@_half_entry
def fused_indice_conv(features, filters, bias, indice_pairs, indice_pair_num, num_activate_out, inverse, subm):
    """sparse_conv_ext.fused_indice_conv_fp32 (fused_spconv_ops.h:28-132): conv + bias."""
    pairs = indice_pairs.contiguous()
    if inverse:
        pairs = pairs.flip(1).contiguous()
    nbr = _ops.pairs_to_nbr(pairs, indice_pair_num, int(num_activate_out))
    cin = features.shape[1]
    cout = filters.shape[-1]
    return _ops.sparse_conv_fused(features.contiguous(), filters.contiguous().view(-1, cin, cout), nbr,
                                  int(num_activate_out), bias=bias.contiguous())


@_half_entry
def indice_maxpool(features, indice_pairs, indice_pair_num, num_activate_out):
    """sparse_conv_ext.indice_maxpool_fp32 (pool_ops.h:26-58) from a reference-format rulebook."""
    if features.dtype != torch.float32:
        raise Df3dError("indice_maxpool: fp32 only (got %s)" % features.dtype)
    nbr = _ops.pairs_to_nbr(indice_pairs.contiguous(), indice_pair_num, int(num_activate_out))
    return _ops.sparse_maxpool(features.contiguous(), nbr, int(num_activate_out))


@_half_entry
def indice_maxpool_backward(features, out_features, out_bp, indice_pairs, indice_pair_num):
    """sparse_conv_ext.indice_maxpool_backward_fp32 (pool_ops.h:60-94)."""
    if features.dtype != torch.float32:
        raise Df3dError("indice_maxpool_backward: fp32 only (got %s)" % features.dtype)
    nbr = _ops.pairs_to_nbr(indice_pairs.contiguous(), indice_pair_num, int(out_features.shape[0]))
    inv = _ops.invert_neighbors(nbr, features.shape[0])
    return _ops.sparse_maxpool_backward(features.contiguous(), out_features.contiguous(), out_bp.contiguous(), inv)
