"""SparseModule / SparseSequential with the reference's container semantics
(TF/mmdet3d/ops/spconv/modules.py:43-137).  In eval mode the container folds
conv -> BatchNorm1d -> ReLU runs into the conv kernel's epilogue (the reference anticipates
this with `fused()` / fused_indice_conv, modules.py:139-187, but never enables it)."""
import sys
from collections import OrderedDict

import torch
from torch import nn

from .structure import SparseConvTensor


def is_spconv_module(module):
    return isinstance(module, (SparseModule,))


def is_sparse_conv(module):
    from .conv import SparseConvolution
    return isinstance(module, SparseConvolution)


class SparseModule(nn.Module):
    """place holder: every subclass takes a SparseConvTensor inside SparseSequential."""
    pass


_FOLD_CACHE = {}


def fold_batchnorm(bn):
    """(scale, shift) of an eval-mode BatchNorm1d: y = x * scale + shift.  Cached on the
    parameter versions so a steady-state forward launches nothing for it."""
    key = id(bn)
    ver = (bn.weight._version if bn.weight is not None else -1, bn.bias._version if bn.bias is not None else -1,
           bn.running_mean._version, bn.running_var._version, bn.running_mean.data_ptr(), bn.eps)
    hit = _FOLD_CACHE.get(key)
    if hit is not None and hit[0] == ver:
        return hit[1], hit[2]
    with torch.no_grad():
        inv = torch.rsqrt(bn.running_var.float() + bn.eps)
        w = bn.weight.float() if bn.weight is not None else torch.ones_like(inv)
        b = bn.bias.float() if bn.bias is not None else torch.zeros_like(inv)
        scale = (w * inv).contiguous()
        shift = (b - bn.running_mean.float() * scale).contiguous()
    _FOLD_CACHE[key] = (ver, scale, shift)
    return scale, shift


def can_fold(bn):
    return isinstance(bn, nn.BatchNorm1d) and (not bn.training) and bn.track_running_stats \
        and bn.running_mean is not None


class SparseSequential(SparseModule):
    def __init__(self, *args, **kwargs):
        super(SparseSequential, self).__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if sys.version_info < (3, 6):
                raise ValueError('kwargs only supported in py36+')
            if name in self._modules:
                raise ValueError('name exists.')
            self.add_module(name, module)
        self._sparity_dict = {}

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError('index {} is out of range'.format(idx))
        if idx < 0:
            idx += len(self)
        it = iter(self._modules.values())
        for i in range(idx):
            next(it)
        return next(it)

    def __len__(self):
        return len(self._modules)

    @property
    def sparity_dict(self):
        return self._sparity_dict

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError('name exists')
        self.add_module(name, module)

    def forward(self, input):
        mods = list(self._modules.items())
        i = 0
        while i < len(mods):
            k, module = mods[i]
            if is_sparse_conv(module) and isinstance(input, SparseConvTensor) and not module.training \
                    and i + 1 < len(mods) and can_fold(mods[i + 1][1]) and not module.conv1x1:
                # conv -> BN(eval) [-> ReLU] in one kernel
                scale, shift = fold_batchnorm(mods[i + 1][1])
                relu = i + 2 < len(mods) and isinstance(mods[i + 2][1], nn.ReLU)
                self._sparity_dict[k] = input.sparity
                input = module.forward_fused(input, scale=scale, shift=shift, relu=relu)
                i += 3 if relu else 2
                continue
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                self._sparity_dict[k] = input.sparity
                input = module(input)
            else:
                if isinstance(input, SparseConvTensor):
                    if input.indices.shape[0] != 0:
                        input.features = module(input.features)
                else:
                    input = module(input)
            i += 1
        return input

    def fused(self):
        """Reference API (modules.py:139-187).  Fusion happens automatically in eval mode here."""
        return self


class ToDense(SparseModule):
    def forward(self, x):
        return x.dense()


class RemoveGrid(SparseModule):
    def forward(self, x):
        x.grid = None
        return x
