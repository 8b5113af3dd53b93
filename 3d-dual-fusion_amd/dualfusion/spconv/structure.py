"""SparseConvTensor: same container API as the reference's spconv
(TF/mmdet3d/ops/spconv/structure.py:21-69; spconv 2.x `replace_feature` as probed by
CP/det3d/models/backbones/scn.py:17-23), backed by the MI355X kernels."""
import os

import numpy as np
import torch

from .. import ops as _ops
from .._lib import Df3dError


def scatter_nd(indices, updates, shape):
    """TF/mmdet3d/ops/spconv/structure.py:5-18 (pure tensor op, kept for API parity)."""
    ret = torch.zeros(*shape, dtype=updates.dtype, device=updates.device)
    ndim = indices.shape[-1]
    output_shape = list(indices.shape[:-1]) + shape[indices.shape[-1]:]
    flatted_indices = indices.view(-1, ndim)
    slices = [flatted_indices[:, i] for i in range(ndim)]
    slices += [Ellipsis]
    ret[slices] = updates.view(*output_shape)
    return ret


class Rulebook(object):
    """What `indice_dict[key]` holds.  Unpacks like the reference's 5-tuple
    (outids, indices, indice_pairs, indice_pair_num, spatial_shape) (conv.py:169-172); the
    reference-format pairs are derived lazily from the neighbour table the kernels use."""

    def __init__(self, outids, indices, nbr, spatial_shape, out_spatial_shape, out_rows_sorted):
        self.outids, self.indices, self.nbr = outids, indices, nbr
        self.spatial_shape, self.out_spatial_shape = spatial_shape, out_spatial_shape
        self.out_rows_sorted = out_rows_sorted
        self._pairs = None
        self._tiles = {}

    def tiles(self, cin, cout):
        """Pair-balanced row ranges of this rulebook for a (cin, cout) layer; computed once, shared by
        every conv that reuses the rulebook."""
        key = (int(cin), int(cout))
        if key not in self._tiles:
            self._tiles[key] = _ops.conv_tiles(self.nbr, cin, cout)
        return self._tiles[key]

    def pairs(self):
        if self._pairs is None:
            self._pairs = _ops.nbr_to_pairs(self.nbr, self.indices.shape[0])
        return self._pairs

    def _tuple(self):
        p, n = self.pairs()
        return (self.outids, self.indices, p, n, self.spatial_shape)

    def __iter__(self):
        return iter(self._tuple())

    def __getitem__(self, i):
        return self._tuple()[i]

    def __len__(self):
        return 5


class _DenseFunction(torch.autograd.Function):
    """dense() with a backward (structure.py:5-18 scatter_nd: the gradient is the gather at the active sites)."""

    @staticmethod
    def forward(ctx, features, indices, batch_size, spatial_shape):
        ctx.save_for_backward(indices)
        return _ops.sparse_to_dense(features, indices, batch_size, list(spatial_shape))

    @staticmethod
    def backward(ctx, grad):
        (indices,) = ctx.saved_tensors
        idx = indices.long()
        g = grad[(idx[:, 0], slice(None)) + tuple(idx[:, i + 1] for i in range(idx.shape[1] - 1))]
        return g.contiguous(), None, None, None


class DirectoryCache(object):
    """Occupancy directories of the index sets of one frame, found again by the index tensor they belong to.
    An entry PINS its index tensor (strong reference): the storage cannot be recycled for another tensor while the
    entry exists, so the (address, rows) key is unambiguous, and the tensor's version counter catches in-place
    edits of the coordinates."""

    def __init__(self):
        self._entries = {}

    def get(self, indices):
        hit = self._entries.get((indices.data_ptr(), indices.shape[0]))
        if hit is not None and hit[1] == hit[0]._version == indices._version:
            return hit[2]
        return None

    def put(self, indices, directory):
        self._entries[(indices.data_ptr(), indices.shape[0])] = (indices, indices._version, directory)

    def __len__(self):
        return len(self._entries)


class SparseConvTensor(object):
    def __init__(self, features, indices, spatial_shape, batch_size, grid=None):
        self.features = features
        self.indices = indices if indices.dtype == torch.int32 else indices.int()
        self.spatial_shape = [int(s) for s in spatial_shape]
        self.batch_size = int(batch_size)
        self.indice_dict = {}
        self.grid = grid
        # occupancy directories keyed by the identity of the indices tensor they index; shared by
        # reference with derived tensors exactly like indice_dict
        self._directories = DirectoryCache()
        # (features tensor, its split rows) when a split-precision conv produced or consumed these features
        self._split = None
        # rulebooks built ahead of time for specific conv modules (dualfusion/executor.py), keyed by id(module)
        self._prebuilt = {}

    def split_features(self):
        """Split rows (bf16 hi | lo) of `features` for the split-precision conv kernels; emitted by the producing
        conv's epilogue when there is one, otherwise computed here once."""
        if (self._split is None or self._split[0] is not self.features
                or self._split[1].shape[1] != _ops.split_width(self.features.shape[1])):      # (another precision mode's rows)
            feats = self.features.contiguous()
            self._split = (self.features, _ops.split_rows(feats))
        return self._split[1]

    def bf16_features(self):
        """`features` as bf16 rows for the bf16 conv kernels (DF3D_CONV_PRECISION=bf16); emitted by the producing
        conv's epilogue when there is one, otherwise converted here once."""
        hit = getattr(self, "_bf16", None)
        if hit is None or hit[0] is not self.features:
            hit = (self.features, _ops.rows_to_bf16(self.features.contiguous().float()))
            self._bf16 = hit
        return hit[1]

    # ---- 2-D tensors run on the 3-D kernels as a one-slice volume ------------------------------------------------
    def lift3d(self):
        """[N, 3] (b, y, x) indices on [H, W] -> the same tensor as (b, 0, y, x) on [1, H, W]; cached, and shares the
        rulebook / directory dictionaries like every derived tensor."""
        hit = self.__dict__.get("_lift")
        if hit is None or hit[0] is not self.indices:
            ind = self.indices
            ind3 = torch.cat([ind[:, :1], torch.zeros_like(ind[:, :1]), ind[:, 1:]], 1).contiguous()
            t = SparseConvTensor(self.features, ind3, [1] + self.spatial_shape, self.batch_size, self.grid)
            t.indice_dict, t._directories = self.indice_dict, self._directories
            hit = (self.indices, t)
            self.__dict__["_lift"] = hit
        t = hit[1]
        if t.features is not self.features:
            t.features, t._split = self.features, self._split
        return t

    def drop_z(self):
        """inverse of lift3d for a tensor produced by a (1, kh, kw) kernel."""
        ind2 = self.indices[:, [0, 2, 3]].contiguous()
        t = SparseConvTensor(self.features, ind2, self.spatial_shape[1:], self.batch_size, self.grid)
        t.indice_dict, t._directories, t._split = self.indice_dict, self._directories, self._split
        t.__dict__["_lift"] = (ind2, self)
        return t

    # ---- reference API ------------------------------------------------------------
    @property
    def spatial_size(self):
        return np.prod(self.spatial_shape)

    def find_indice_pair(self, key):
        if key is None:
            return None
        return self.indice_dict.get(key, None)

    def dense(self, channels_first=True):
        """[B, C, *spatial] (channels_first) like structure.py:55-64; one fused kernel."""
        if len(self.spatial_shape) == 2:                     # one-slice volume, z dropped again
            return self.lift3d().dense(channels_first).squeeze(2 if channels_first else 1)
        if len(self.spatial_shape) != 3 or self.indices.shape[1] != 4:
            raise Df3dError("dense(): 2-D or 3-D tensors with [N, 1 + ndim] indices only")
        feats = self.features.contiguous()
        if feats.is_cuda and os.environ.get("DF3D_STRICT", "0") == "1":
            _ops.check_split_overflow()                       # strict mode: synchronise and raise on a raised range flag
        if feats.requires_grad and torch.is_grad_enabled():
            out = _DenseFunction.apply(feats, self.indices.contiguous(), self.batch_size, tuple(self.spatial_shape))
        else:
            out = _ops.sparse_to_dense(feats, self.indices.contiguous(), self.batch_size, self.spatial_shape)
        if not channels_first:
            nd = len(self.spatial_shape)
            return out.permute(0, *range(2, nd + 2), 1).contiguous()
        return out

    def dense_rows(self, split=False):
        """dense().view(B, C*D, H, W) as channels-last pixel rows [B*H*W, C*D] (3-D tensors only).
        split: the same rows in the operand format of the split-precision kernels only ([B*H*W, 4 * C*D] uint8)."""
        if split:
            return _ops.sparse_to_dense_rows_split(self.features.contiguous(), self.indices.contiguous(), self.batch_size,
                                                   self.spatial_shape)
        return _ops.sparse_to_dense_rows(self.features.contiguous(), self.indices.contiguous(), self.batch_size,
                                         self.spatial_shape)

    @property
    def sparity(self):
        return self.indices.shape[0] / np.prod(self.spatial_shape) / self.batch_size

    def replace_feature(self, new_features):
        """spconv 2.x style functional update (scn.py:17-23 probes for it)."""
        out = SparseConvTensor(new_features, self.indices, self.spatial_shape, self.batch_size, self.grid)
        out.indice_dict = self.indice_dict
        out._directories = self._directories
        out._prebuilt = self._prebuilt
        out._indices_synced = getattr(self, "_indices_synced", False)
        return out

    # ---- directory cache ----------------------------------------------------------
    def directory(self, rows_sorted=False):
        d = self._directories.get(self.indices)
        if d is None:
            d = _ops.grid_build(self.indices.contiguous(), self.batch_size, self.spatial_shape, rows_sorted=rows_sorted)
            self._directories.put(self.indices, d)
        return d
