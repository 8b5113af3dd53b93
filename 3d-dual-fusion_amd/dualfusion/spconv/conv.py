"""Sparse convolution modules with the reference's constructor arguments, parameter names and
shapes (weight [kD,kH,kW,Cin,Cout], bias [Cout]; TF/mmdet3d/ops/spconv/conv.py:48-204) so the
reference's configs and checkpoints load unchanged.  forward() = rulebook (cached per
indice_key, conv.py:146-172) + ONE fused kernel (csrc/spconv.hip)."""
import math

import numpy as np
import torch
from torch.nn import init
from torch.nn.parameter import Parameter

from .. import ops as _ops
from .._lib import Df3dError
from . import ops
from .modules import SparseModule
from .structure import Rulebook, SparseConvTensor


def _fan_in_hwio(weight):
    """fan-in of a filter bank stored [*kernel, Cin, Cout] (the reference keeps the channels LAST, so torch's own
    fan computation -- which assumes [Cout, Cin, *kernel] -- would be wrong): taps x input channels."""
    if weight.dim() < 2:
        raise ValueError('fan in and fan out can not be computed for tensor with fewer than 2 dimensions')
    return int(np.prod(weight.shape[:-1]))



# indice_dict entry (shared by every tensor derived from the network input) that holds the stream rulebooks are built on
GEOMETRY_STREAM_KEY = "__df3d_geometry_stream"

class SparseConvFunction(torch.autograd.Function):
    """indice_conv + its backward (TF/mmdet3d/ops/spconv/functional.py:20-75 -> ops.indice_conv /
    ops.indice_conv_backward) on the kernel-facing rulebook: forward = the exact-fp32 fused kernel (conv + bias),
    backward = input gradient by the same kernel on the inverse table, filter gradient by
    df3d_sparse_conv_grad_filters, bias gradient = column sums."""

    @staticmethod
    def forward(ctx, features, filters, bias, nbr, n_out, mirror, inv=None):
        """`inv`: the inverse table [K, n_in] when the caller keeps one (dense layers reuse theirs from step to step);
        otherwise it is the mirrored forward table (`mirror`) or built in backward."""
        K = nbr.shape[0]
        cin, cout = features.shape[1], filters.shape[-1]
        ctx.save_for_backward(features, filters, nbr)
        ctx.mirror, ctx.has_bias, ctx.inv = mirror, bias is not None, inv
        w = filters.detach().contiguous().view(K, cin, cout)
        b = bias.detach() if bias is not None else None
        ctx.mode = _ops.CONV_PRECISION                     # the backward runs in the arithmetic the forward was called in
        ctx.amp = _ops.CONV_PRECISION == "bf16" and _ops.conv_bf16_supported(K, cin, cout)
        if ctx.amp:
            # bf16 mixed-precision training (the reference's fp16-AMP configurations): operands rounded to
            # bf16 for the matrix cores, fp32 accumulate and fp32 rows out (BatchNorm, residuals, losses and the master
            # weights stay fp32, as under torch.autocast); the backward does the same for the input gradient
            return _ops.sparse_conv_bf16(_ops.rows_to_bf16(features), _ops.conv_pack_weights_bf16(w), nbr, n_out, cin, cout,
                                         bias=b, want_f32=True, want_bf16=False)[0]
        if _ops.conv_split_supported(K, cin, cout):            # same split-precision kernel as inference (~1e-5 rel.)
            return _ops.sparse_conv_split(_ops.split_rows(features), _ops.conv_pack_weights(w), nbr, n_out, cin, cout,
                                          bias=b, emit_split=False)[0]
        if _ops.CONV_PRECISION == "split" and cin > 128 and cin % 128 == 0 and cout % 8 == 0 and cout <= 128:
            # many input channels (a head's shared convolution 512 -> 128 over the BEV rows): the row kernel that walks the
            # input channels in 128-column blocks (the inference path's launch, csrc/spconv_split.hip loader / consumer)
            return _ops.conv_rows_split(_ops.split_rows(features), cin, 0, _ops.conv_pack_weights(w), cout, 1, nbr, n_out, b)[0]
        return _ops.sparse_conv_fused(features, w, nbr, n_out, bias=b)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        features, filters, nbr = ctx.saved_tensors
        K = nbr.shape[0]
        cin, cout = features.shape[1], filters.shape[-1]
        with _ops.precision(ctx.mode):
            g_in, g_w = _ops.sparse_conv_backward(features, filters.detach().contiguous().view(K, cin, cout),
                                                  grad_out.contiguous().float(), nbr, ctx.mirror, inv=ctx.inv, bf16=ctx.amp)
        g_b = grad_out.sum(0) if ctx.has_bias else None
        return g_in, g_w.view_as(filters), g_b, None, None, None, None


class SparseConvolution(SparseModule):
    def __init__(self, ndim, in_channels, out_channels, kernel_size=3, stride=1, padding=0, dilation=1, groups=1,
                 bias=True, subm=False, output_padding=0, transposed=False, inverse=False, indice_key=None,
                 fused_bn=False):
        super(SparseConvolution, self).__init__()
        assert groups == 1
        lst = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * ndim
        kernel_size, stride, padding, dilation = lst(kernel_size), lst(stride), lst(padding), lst(dilation)
        output_padding = lst(output_padding)
        for d, s in zip(dilation, stride):
            assert any([s == 1, d == 1]), "don't support this."
        self.ndim = ndim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.kernel_size = kernel_size
        self.conv1x1 = np.prod(kernel_size) == 1
        self.stride = stride
        self.padding = padding
        self.dilation = dilation
        self.transposed = transposed
        self.inverse = inverse
        self.output_padding = output_padding
        self.groups = groups
        self.subm = subm
        self.indice_key = indice_key
        self.fused_bn = fused_bn
        self.weight = Parameter(torch.Tensor(*kernel_size, in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        """Same distributions as the reference's initialiser (conv.py:106-112): Kaiming-uniform filters with
        a = sqrt(5) (fan counted by torch on the stored layout, as the reference does) and a bias uniform in
        +-1/sqrt(taps * Cin)."""
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            lim = 1.0 / math.sqrt(_fan_in_hwio(self.weight))
            with torch.no_grad():
                self.bias.uniform_(-lim, lim)

    # ------------------------------------------------------------------------------
    def _rulebook(self, input):
        """Reuse the rulebook of `indice_key` (conv.py:146-155) or build and register it."""
        pre = input._prebuilt.get(id(self)) if input._prebuilt else None
        if pre is not None and pre.indices is input.indices:
            return pre
        datas = input.find_indice_pair(self.indice_key)
        if self.inverse:
            # conv.py:146-151: the rulebook of the forward conv registered under `indice_key`, read backwards -- the
            # outputs are that conv's INPUT sites, every pair (i, o) of offset k contributes in[o] . W[k] to out[i]
            if datas is None or self.indice_key is None:
                raise Df3dError("SparseInverseConv needs the rulebook of a previous convolution with the same indice_key")
            if datas.nbr.shape[0] != int(np.prod(self.kernel_size)):
                raise Df3dError("SparseInverseConv: kernel size differs from the convolution registered as %r" % (self.indice_key,))
            inv = getattr(datas, "_inverse", None)
            if inv is None:
                inv = Rulebook(datas.indices, datas.outids, _ops.invert_neighbors(datas.nbr, datas.indices.shape[0]),
                               datas.out_spatial_shape, datas.spatial_shape, out_rows_sorted=False)
                datas._inverse = inv
            return inv
        if self.indice_key is not None and datas is not None:
            return datas
        # SubM convs without an indice_key (every BasicBlock conv of the TransFusion encoder,
        # sparse_block.py:85-100) rebuild an identical rulebook in the reference; the neighbour table
        # only depends on the index set and the kernel geometry, so it is shared here.
        auto_key = None
        if self.subm and self.indice_key is None:
            auto_key = ("__subm", tuple(self.kernel_size), tuple(self.dilation), input.indices.data_ptr(),
                        input.indices.shape[0])
            hit = input.indice_dict.get(auto_key)
            if hit is not None:
                return hit
        geo = input.indice_dict.get(GEOMETRY_STREAM_KEY)
        if geo is not None and input.indices.is_cuda:
            # Training on resident inputs (backbones._stem): the rulebooks depend on the coordinates alone, so they are built
            # on a side stream that was ordered behind the voxeliser's event only -- the host's round trip for an output
            # count then waits for the few geometry kernels in front of it, not for the previous step's backward still
            # queued on the caller's stream.  The caller's stream waits for the side stream; every tensor produced there is
            # marked as used on the caller's stream (the allocator must not hand it to the next step's geometry early).
            main = torch.cuda.current_stream(input.indices.device)
            with torch.cuda.stream(geo):
                directory = input.directory()
                outids, nbr, out_shape, out_dir = ops.build_rulebook(input.indices, input.batch_size, input.spatial_shape,
                                                                     self.kernel_size, self.stride, self.padding,
                                                                     self.dilation, self.subm, directory=directory,
                                                                     transpose=self.transposed,
                                                                     out_padding=self.output_padding)
            main.wait_stream(geo)
            for d in (directory, out_dir):
                for t in ((d.blob, d.perm) if d is not None else ()):
                    if t is not None:
                        t.record_stream(main)
            for t in (outids, nbr):
                t.record_stream(main)
        else:
            directory = input.directory()
            outids, nbr, out_shape, out_dir = ops.build_rulebook(input.indices, input.batch_size, input.spatial_shape,
                                                                 self.kernel_size, self.stride, self.padding,
                                                                 self.dilation, self.subm, directory=directory,
                                                                 transpose=self.transposed, out_padding=self.output_padding)
        rb = Rulebook(outids, input.indices, nbr, input.spatial_shape, out_shape, out_rows_sorted=not self.subm)
        if out_dir is not None and not self.subm:
            input._directories.put(outids, out_dir)
        input.indice_dict[auto_key if auto_key is not None else self.indice_key] = rb
        return rb

    def forward_fused(self, input, scale=None, shift=None, relu=False, residual=None):
        """out = act((conv(x) + bias) * scale + shift + residual): the conv kernel's epilogue.  `residual`: fp32 rows
        or the SparseConvTensor that holds them (lets the bf16 mode reuse its bf16 rows)."""
        assert isinstance(input, SparseConvTensor)
        res_sct = residual if isinstance(residual, SparseConvTensor) else None
        if res_sct is not None:
            residual = res_sct.features.contiguous()
        if self.ndim == 2:
            # SparseConv2d / SubMConv2d: the same kernels on a one-slice volume, kernel (1, kh, kw)
            return SparseConvolution.forward_fused(self._twin3d(), input.lift3d(), scale, shift, relu, residual).drop_z()
        if self.ndim != 3:
            raise Df3dError("only 2-D and 3-D sparse convolutions are implemented on the MI355X path")
        if self.conv1x1:
            feats = torch.mm(input.features, self.weight.view(self.in_channels, self.out_channels))
            if self.bias is not None:
                feats = feats + self.bias
            if scale is not None:
                feats = feats * scale + shift
            if residual is not None:
                feats = feats + residual
            if relu:
                feats = torch.relu(feats)
            out = SparseConvTensor(feats, input.indices, input.spatial_shape, input.batch_size)
            out.indice_dict, out.grid, out._directories = input.indice_dict, input.grid, input._directories
            return out
        rb = self._rulebook(input)
        feats = input.features
        if feats.dtype != torch.float32:
            feats = feats.float()
        K = rb.nbr.shape[0]
        n_out = rb.outids.shape[0]
        if torch.is_grad_enabled() and (self.weight.requires_grad or feats.requires_grad):
            # training: plain convolution (+ bias) through the autograd Function; BatchNorm / ReLU / residual stay
            # separate modules exactly as in the reference (conv.py:179-204, functional.py:20-75)
            if scale is not None or shift is not None or residual is not None or relu:
                raise Df3dError("the fused BatchNorm / ReLU / residual epilogue is inference-only; call the conv alone "
                                "when gradients are required")
            out_features = SparseConvFunction.apply(feats.contiguous(), self.weight, self.bias, rb.nbr, n_out,
                                                    bool(self.subm and all(k % 2 == 1 for k in self.kernel_size)))
            out = SparseConvTensor(out_features, rb.outids, rb.out_spatial_shape, input.batch_size)
            out.indice_dict, out.grid, out._directories = input.indice_dict, input.grid, input._directories
            return out
        w = self.weight.detach()
        bias = self.bias.detach() if self.bias is not None else None
        tiles = rb.tiles(self.in_channels, self.out_channels)
        out_split = None
        if _ops.CONV_PRECISION == "bf16" and _ops.conv_bf16_supported(K, self.in_channels, self.out_channels):
            # BASELINE configs[2]: bf16 rows and weights, fp32 accumulate; the fp32 copy of the result serves the
            # layers without a bf16 kernel (C <= 16) and the fusion adapter
            if input.features is not feats:
                input = input.replace_feature(feats)
            if residual is None:
                res16 = None
            elif res_sct is not None:
                res16 = res_sct.bf16_features()
            else:
                res16 = _ops.rows_to_bf16(residual.contiguous().float())
            out_features, out16 = _ops.sparse_conv_bf16(input.bf16_features(), self._packed_weight_bf16(w, K), rb.nbr,
                                                        n_out, self.in_channels, self.out_channels, bias=bias,
                                                        scale=scale, shift=shift, residual=res16, relu=relu,
                                                        want_f32=True, want_bf16=True)
            out = SparseConvTensor(out_features, rb.outids, rb.out_spatial_shape, input.batch_size)
            out.indice_dict, out.grid, out._directories = input.indice_dict, input.grid, input._directories
            out._bf16 = (out_features, out16)
            return out
        if _ops.conv_split_supported(K, self.in_channels, self.out_channels):
            if input.features is not feats:
                input = input.replace_feature(feats)
            out_features, out_split = _ops.sparse_conv_split(
                input.split_features(), self._packed_weight(w, K), rb.nbr, n_out, self.in_channels,
                self.out_channels, bias=bias, scale=scale, shift=shift, residual=residual, relu=relu, tiles=tiles)
        else:
            out_features = _ops.sparse_conv_fused(feats.contiguous(),
                                                  w.contiguous().view(K, self.in_channels, self.out_channels),
                                                  rb.nbr, n_out, bias=bias, scale=scale, shift=shift,
                                                  residual=residual, relu=relu, tiles=tiles)
        out = SparseConvTensor(out_features, rb.outids, rb.out_spatial_shape, input.batch_size)
        out.indice_dict, out.grid, out._directories = input.indice_dict, input.grid, input._directories
        if out_split is not None:
            out._split = (out_features, out_split)
        return out

    def _twin3d(self):
        """This 2-D module as a 3-D one (kernel (1, kh, kw), ...) sharing its parameters."""
        twin = self.__dict__.get("_twin")
        if twin is None:
            import copy
            twin = copy.copy(self)
            twin.__dict__.pop("_twin", None)
            twin.ndim = 3
            twin.kernel_size, twin.stride = [1] + list(self.kernel_size), [1] + list(self.stride)
            twin.padding, twin.dilation = [0] + list(self.padding), [1] + list(self.dilation)
            twin.output_padding = [0] + list(self.output_padding) if isinstance(self.output_padding, (list, tuple)) else self.output_padding
            self.__dict__["_twin"] = twin
        return twin

    def _packed_weight_bf16(self, w, K):
        key = (w.data_ptr(), w._version, str(w.device))
        hit = getattr(self, "_packed16", None)
        if hit is None or hit[0] != key:
            hit = (key, _ops.conv_pack_weights_bf16(w.contiguous().float().view(K, self.in_channels, self.out_channels)))
            self._packed16 = hit
        return hit[1]

    def _packed_weight(self, w, K):
        """hi/lo bf16 MFMA operands of the filter bank, rebuilt only when the parameter changes."""
        key = (w.data_ptr(), w._version, str(w.device), _ops.split_parts())
        hit = getattr(self, "_packed", None)
        if hit is None or hit[0] != key:
            hit = (key, _ops.conv_pack_weights(w.contiguous().view(K, self.in_channels, self.out_channels)))
            self._packed = hit
        return hit[1]

    def forward(self, input):
        return self.forward_fused(input)


class SparseConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super(SparseConv2d, self).__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation,
                                           groups, bias, indice_key=indice_key)


class SparseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super(SparseConv3d, self).__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                                           groups, bias, indice_key=indice_key)


class SparseConvTranspose2d(SparseConvolution):
    """conv.py:262-285 of the vendored spconv."""
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super(SparseConvTranspose2d, self).__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation,
                                                    groups, bias, transposed=True, indice_key=indice_key)


class SparseConvTranspose3d(SparseConvolution):
    """conv.py:287-310 of the vendored spconv: every input voxel writes the cells in*stride - pad + c*dilation; the
    same gather-GEMM kernels on the transposed rulebook."""
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super(SparseConvTranspose3d, self).__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation,
                                                    groups, bias, transposed=True, indice_key=indice_key)


class SubMConv2d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super(SubMConv2d, self).__init__(2, in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                                         bias, True, indice_key=indice_key)


class SubMConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 indice_key=None):
        super(SubMConv3d, self).__init__(3, in_channels, out_channels, kernel_size, stride, padding, dilation, groups,
                                         bias, True, indice_key=indice_key)


class SparseInverseConv3d(SparseConvolution):
    def __init__(self, in_channels, out_channels, kernel_size, indice_key, bias=True):
        super(SparseInverseConv3d, self).__init__(3, in_channels, out_channels, kernel_size, bias=bias, inverse=True,
                                                  indice_key=indice_key)
