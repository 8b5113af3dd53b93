"""`spconv`-shaped namespace (the CenterPoint / OpenPCDet trees use it as `spconv.X`,
CP/det3d/models/backbones/scn.py:2-9; the TransFusion tree vendors it as mmdet3d.ops.spconv)."""
from . import ops
from .conv import (SparseConv2d, SparseConv3d, SparseConvolution, SparseConvTranspose2d, SparseConvTranspose3d,
                   SparseInverseConv3d, SubMConv2d, SubMConv3d)
from .pool import SparseMaxPool2d, SparseMaxPool3d
from .modules import RemoveGrid, SparseModule, SparseSequential, ToDense
from .structure import Rulebook, SparseConvTensor, scatter_nd

__all__ = ['SparseConv2d', 'SparseConv3d', 'SubMConv2d', 'SubMConv3d', 'SparseInverseConv3d', 'SparseModule',
           'SparseSequential', 'SparseConvTensor', 'scatter_nd', 'ToDense', 'RemoveGrid', 'ops', 'SparseConvolution',
           'Rulebook', 'SparseMaxPool2d', 'SparseMaxPool3d', 'SparseConvTranspose2d', 'SparseConvTranspose3d']
