"""SparseMaxPool2d / SparseMaxPool3d with the reference's constructor arguments (TF/mmdet3d/ops/spconv/pool.py:21-85;
autograd wrapper functional.py:77-92) on `df3d_sparse_maxpool`: one launch over the neighbour table per direction
instead of one per kernel offset.  Like the reference the pooled value starts from zero (max(0, inputs))."""
import torch

from .. import ops as _ops
from .._lib import Df3dError
from . import ops
from .modules import SparseModule
from .structure import SparseConvTensor


class SparseMaxPoolFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, nbr, n_out):
        out = _ops.sparse_maxpool(features.contiguous(), nbr, n_out)
        ctx.save_for_backward(nbr, features, out)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_output):
        nbr, features, out = ctx.saved_tensors
        inv = _ops.invert_neighbors(nbr, features.shape[0])
        return _ops.sparse_maxpool_backward(features.contiguous(), out, grad_output.contiguous(), inv), None, None


class SparseMaxPool(SparseModule):
    def __init__(self, ndim, kernel_size, stride=1, padding=0, dilation=1, subm=False):
        super(SparseMaxPool, self).__init__()
        lst = lambda v: list(v) if isinstance(v, (list, tuple)) else [v] * ndim
        self.ndim = ndim
        self.kernel_size, self.stride, self.padding, self.dilation = lst(kernel_size), lst(stride), lst(padding), lst(dilation)
        self.subm = subm

    def forward(self, input):
        assert isinstance(input, SparseConvTensor)
        if self.ndim == 2:                                  # one-slice volume, kernel (1, kh, kw)
            twin = self.__dict__.get("_twin")
            if twin is None:
                twin = self.__dict__["_twin"] = SparseMaxPool(3, [1] + self.kernel_size, [1] + self.stride,
                                                              [0] + self.padding, [1] + self.dilation, self.subm)
            return twin(input.lift3d()).drop_z()
        if self.ndim != 3:
            raise Df3dError("only 2-D and 3-D sparse max pooling is implemented on the MI355X path")
        feats = input.features
        if feats.dtype != torch.float32:
            raise Df3dError("SparseMaxPool: fp32 only (got %s)" % feats.dtype)
        outids, nbr, out_shape, out_dir = ops.build_rulebook(input.indices, input.batch_size, input.spatial_shape,
                                                             self.kernel_size, self.stride, self.padding, self.dilation,
                                                             self.subm, directory=input.directory())
        out = SparseConvTensor(SparseMaxPoolFunction.apply(feats, nbr, outids.shape[0]), outids, out_shape, input.batch_size)
        out.indice_dict, out.grid, out._directories = input.indice_dict, input.grid, input._directories
        if out_dir is not None and not self.subm:
            input._directories.put(outids, out_dir)
        return out


class SparseMaxPool2d(SparseMaxPool):
    def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
        super(SparseMaxPool2d, self).__init__(2, kernel_size, stride, padding, dilation)


class SparseMaxPool3d(SparseMaxPool):
    def __init__(self, kernel_size, stride=1, padding=0, dilation=1):
        super(SparseMaxPool3d, self).__init__(3, kernel_size, stride, padding, dilation)
