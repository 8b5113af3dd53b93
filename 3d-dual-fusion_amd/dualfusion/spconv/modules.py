"""SparseModule / SparseSequential with the reference's container semantics
(TF/mmdet3d/ops/spconv/modules.py:43-137).  In eval mode the container folds
conv -> BatchNorm1d -> ReLU runs into the conv kernel's epilogue (the reference anticipates
this with `fused()` / fused_indice_conv, modules.py:139-187, but never enables it)."""
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


def fold_batchnorm(bn):
    """(scale, shift) of an eval-mode BatchNorm1d: y = x * scale + shift.  The folded pair lives ON the module
    (so it dies with it and can never be handed to another module) and is rebuilt when any parameter / buffer
    changes version, storage or device -- a steady-state forward launches nothing for it."""
    tensors = (bn.weight, bn.bias, bn.running_mean, bn.running_var)
    ver = tuple((-1, 0) if t is None else (t._version, t.data_ptr()) for t in tensors) + (bn.eps,)
    hit = bn.__dict__.get("_df3d_fold")
    if hit is not None and hit[0] == ver:
        return hit[1], hit[2]
    with torch.no_grad():
        inv = torch.rsqrt(bn.running_var.float() + bn.eps)
        w = bn.weight.float() if bn.weight is not None else torch.ones_like(inv)
        b = bn.bias.float() if bn.bias is not None else torch.zeros_like(inv)
        scale = (w * inv).contiguous()
        shift = (b - bn.running_mean.float() * scale).contiguous()
    bn.__dict__["_df3d_fold"] = (ver, scale, shift)
    return scale, shift


def can_fold(bn):
    return isinstance(bn, nn.BatchNorm1d) and (not bn.training) and bn.track_running_stats \
        and bn.running_mean is not None


def wants_grad(x, *modules):
    """True when autograd would have to record this call: grad mode on and the input rows or any parameter of
    `modules` require grad.  The fused conv + BatchNorm + ReLU epilogue is inference-only, so callers take the
    unfused module sequence (the reference's composition) in that case -- e.g. `model.eval()` without
    `torch.no_grad()`, frozen-BN fine-tuning, gradient-based analysis."""
    if not torch.is_grad_enabled():
        return False
    feats = getattr(x, "features", x)
    if isinstance(feats, torch.Tensor) and feats.requires_grad:
        return True
    return any(p.requires_grad for m in modules if m is not None for p in m.parameters())


class SparseSequential(SparseModule):
    """Ordered container of sparse / dense modules.  Construction forms of the reference
    (TF/mmdet3d/ops/spconv/modules.py:43-137): positional modules (named "0", "1", ...), one OrderedDict, and
    keyword modules appended after either; `seq[i]`, `len(seq)`, `seq.add(module, name)`, `seq.sparity_dict`."""

    def __init__(self, *args, **kwargs):
        super(SparseSequential, self).__init__()
        self._sparity_dict = {}
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            entries = list(args[0].items())
        else:
            entries = [(str(pos), mod) for pos, mod in enumerate(args)]
        for name, mod in entries:
            self.add_module(name, mod)
        for name, mod in kwargs.items():
            if name in self._modules:
                raise ValueError('name exists.')
            self.add_module(name, mod)

    def __len__(self):
        return len(self._modules)

    def __getitem__(self, idx):
        members = tuple(self._modules.values())
        if idx < -len(members) or idx >= len(members):
            raise IndexError('index {} is out of range'.format(idx))
        return members[idx]

    @property
    def sparity_dict(self):
        return self._sparity_dict

    def add(self, module, name=None):
        """Append `module`; unnamed modules take their position as name."""
        if name is None:
            name = str(len(self))
            if name in self._modules:
                raise KeyError('name exists')
        self.add_module(name, module)

    def forward(self, input):
        mods = list(self._modules.items())
        i = 0
        while i < len(mods):
            k, module = mods[i]
            if is_sparse_conv(module) and isinstance(input, SparseConvTensor) and not module.training \
                    and i + 1 < len(mods) and can_fold(mods[i + 1][1]) and not module.conv1x1 \
                    and not wants_grad(input, module, mods[i + 1][1]):
                # conv -> BN(eval) [-> ReLU] in one kernel
                scale, shift = fold_batchnorm(mods[i + 1][1])
                relu = i + 2 < len(mods) and isinstance(mods[i + 2][1], nn.ReLU)
                self._sparity_dict[k] = input.sparity
                input = module.forward_fused(input, scale=scale, shift=shift, relu=relu)
                i += 3 if relu else 2
                continue
            if isinstance(module, nn.BatchNorm1d) and module.training and isinstance(input, SparseConvTensor) \
                    and input.indices.shape[0] > 1 and input.features.is_cuda:
                # BatchNorm1d(train) [-> ReLU] over the feature rows on the row kernels (csrc/bnrows.hip)
                from .. import ops as _ops
                relu = i + 1 < len(mods) and isinstance(mods[i + 1][1], nn.ReLU)
                input.features = _ops.batch_norm_rows(module, input.features, relu)
                i += 2 if relu else 1
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
