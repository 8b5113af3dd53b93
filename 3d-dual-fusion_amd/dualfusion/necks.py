"""BEV neck with the reference's constructor arguments and parameter layouts, so the
README checkpoints load:
  `RPN`                 CP/det3d/models/necks/rpn.py:22-163 (blocks.<i> = ZeroPad2d, Conv2d 3x3, BN, ReLU, then n x (Conv2d
                        3x3 pad 1, BN, ReLU); deblocks.<i> = ConvTranspose2d(k = s) | Conv2d, BN, ReLU)
  `SECOND`, `SECONDFPN` TF/mmdet3d/models/backbones/second.py:9-88, TF/mmdet3d/models/necks/second_fpn.py:12-91
                        (pts_backbone / pts_neck of TF/configs/transfusion_nusc_voxel_LC.py:168-183)

`forward(x)` takes / returns NCHW like the reference.  In eval mode on the GPU the layers do not go through MIOpen:
a dense 3x3 convolution over channels-last pixel rows IS the sparse convolution with a full neighbour table, so
every layer (conv + folded BatchNorm + ReLU) is one launch of the split-precision kernel of csrc/spconv_split.hip
(`forward_rows`; 126 GFLOP per nuScenes sweep).  `SparseConvTensor.dense()` can feed it rows directly
(`ops.sparse_to_dense_rows`), which removes the NCHW volume and its permute."""
import os
import numpy as np
import torch
from torch import nn

from . import ops as _ops
from .registry import MM_BACKBONES, MM_NECKS, NECKS


def _bn(norm_cfg, planes):
    cfg = dict(norm_cfg or dict(type="BN", eps=1e-3, momentum=0.01))
    t = cfg.pop("type", "BN")
    if t not in ("BN", "BN2d"):
        raise KeyError("unsupported norm type for the BEV neck: %s" % t)
    cfg.pop("requires_grad", None)
    return nn.BatchNorm2d(planes, **cfg)


def _bn_rows(bn, x, relu=False):
    """BatchNorm2d (+ ReLU) on channels-last pixel rows [P, C]: `ops.batch_norm_rows` (row kernels in train() mode)."""
    return _ops.batch_norm_rows(bn, x, relu)


def _train_stack(groups, rows, B, H, W, tables):
    """A block / deblock of (conv | deconv, BN, ReLU) groups over pixel rows WITH autograd (training rows, SURVEY.md
    section 8f row 4): every convolution is `SparseConvFunction` on the layer's full neighbour table -- forward and
    input gradient on the split-precision kernel, filter gradient on df3d_sparse_conv_grad_filters -- so training
    does not leave the channels-last rows either; BatchNorm over the rows, ReLU.  -> (rows, H, W)."""
    from .spconv.conv import SparseConvFunction
    for conv, bn, relu, pad in groups:
        transposed = isinstance(conv, nn.ConvTranspose2d)
        kh, kw, stride = int(conv.kernel_size[0]), int(conv.kernel_size[1]), int(conv.stride[0])
        key = (B, H, W, kh, kw, stride, pad, transposed)
        if key not in tables:
            tables[key] = _ops.conv2d_neighbors(B, H, W, kh, kw, stride, pad, transposed, rows.device)
        nbr, Ho, Wo = tables[key]
        same = (not transposed) and stride == 1 and Ho == H and Wo == W and kh % 2 == 1 and kw % 2 == 1
        inv = None
        if not same:                                     # strided / transposed layers: inverse table kept with the table
            ikey = key + ("inv",)
            if ikey not in tables:
                tables[ikey] = _ops.invert_neighbors(nbr, rows.shape[0])
            inv = tables[ikey]
        # [cout, cin, kh, kw] (deconv: [cin, cout, s, s]) -> [kh, kw, cin, cout]; autograd carries the gradient back
        w = conv.weight.permute(2, 3, 0, 1) if transposed else conv.weight.permute(2, 3, 1, 0)
        rows = SparseConvFunction.apply(rows.contiguous(), w, conv.bias, nbr, nbr.shape[1], same, inv)
        rows = _bn_rows(bn, rows, relu)
        H, W = Ho, Wo
    return rows, H, W


def _train_rows_ok(x):
    return x.is_cuda and x.dtype == torch.float32


class SplitRows(object):
    """Pixel rows handed over in the operand format of the split-precision kernels only ([rows, 4 * C] uint8: bf16 hi | lo
    per 8 columns) -- what `SparseConvTensor.dense_rows(split=True)` produces and `RPN.forward_rows` accepts."""

    def __init__(self, split, channels):
        self.split, self.channels = split, int(channels)


class _NoRows(object):
    """Stands in for the fp32 rows of a map that exists as split rows only (the row kernels never read them)."""

    def __init__(self, device):
        self.device = device


class _Layer(object):
    """One conv/deconv + BN(eval) + ReLU group prepared for the row kernels."""

    def __init__(self, conv, bn, relu):
        self.conv, self.bn, self.relu = conv, bn, relu
        self.transposed = isinstance(conv, nn.ConvTranspose2d)
        self._key = None

    def prepare(self, pad):
        c = self.conv
        key = (c.weight.data_ptr(), c.weight._version, self.bn.running_var._version, self.bn.weight._version,
               _ops.CONV_PRECISION)
        if key == self._key:
            return
        w = c.weight.detach().float()
        if self.transposed:                  # [cin, cout, s, s] -> [s*s, cin, cout]
            self.filters = w.permute(2, 3, 0, 1).reshape(-1, w.shape[0], w.shape[1]).contiguous()
        else:                                # [cout, cin, kh, kw] -> [kh*kw, cin, cout]
            self.filters = w.permute(2, 3, 1, 0).reshape(-1, w.shape[1], w.shape[0]).contiguous()
        K, cin, cout = self.filters.shape
        self.packed = _ops.conv_pack_weights(self.filters) if _ops.conv_split_supported(K, cin, cout) else None
        self.packed16 = (_ops.conv_pack_weights_bf16(self.filters)
                         if _ops.CONV_PRECISION == "bf16" and _ops.conv_bf16_supported(K, cin, cout) else None)
        inv = torch.rsqrt(self.bn.running_var.float() + self.bn.eps)
        self.scale = (self.bn.weight.float() * inv).contiguous()
        self.shift = (self.bn.bias.float() - self.bn.running_mean.float() * self.scale).contiguous()
        self.bias = c.bias.detach().float().contiguous() if c.bias is not None else None
        self.kh, self.kw = int(c.kernel_size[0]), int(c.kernel_size[1])
        self.stride = int(c.stride[0])
        self.pad = pad
        self._key = key


@NECKS.register_module
class RPN(nn.Module):
    def __init__(self, layer_nums, ds_layer_strides, ds_num_filters, us_layer_strides, us_num_filters,
                 num_input_features, norm_cfg=None, name="rpn", logger=None, **kwargs):
        super(RPN, self).__init__()
        self._layer_strides = list(ds_layer_strides)
        self._num_filters = list(ds_num_filters)
        self._layer_nums = list(layer_nums)
        self._upsample_strides = list(us_layer_strides)
        self._num_upsample_filters = list(us_num_filters)
        self._num_input_features = num_input_features
        self._norm_cfg = norm_cfg if norm_cfg is not None else dict(type="BN", eps=1e-3, momentum=0.01)
        assert len(self._layer_strides) == len(self._layer_nums) == len(self._num_filters)
        assert len(self._num_upsample_filters) == len(self._upsample_strides)
        self._upsample_start_idx = len(self._layer_nums) - len(self._upsample_strides)
        ratios = [self._upsample_strides[i] / np.prod(self._layer_strides[:i + self._upsample_start_idx + 1])
                  for i in range(len(self._upsample_strides))]
        assert all(r == ratios[0] for r in ratios)
        in_filters = [self._num_input_features] + self._num_filters[:-1]
        blocks, deblocks = [], []
        for i, layer_num in enumerate(self._layer_nums):
            planes = self._num_filters[i]
            mods = [nn.ZeroPad2d(1), nn.Conv2d(in_filters[i], planes, 3, stride=self._layer_strides[i], bias=False),
                    _bn(self._norm_cfg, planes), nn.ReLU()]
            for _ in range(layer_num):
                mods += [nn.Conv2d(planes, planes, 3, padding=1, bias=False), _bn(self._norm_cfg, planes), nn.ReLU()]
            blocks.append(nn.Sequential(*mods))
            j = i - self._upsample_start_idx
            if j >= 0:
                stride = self._upsample_strides[j]
                up = self._num_upsample_filters[j]
                if stride > 1:
                    conv = nn.ConvTranspose2d(planes, up, int(stride), stride=int(stride), bias=False)
                else:
                    st = int(np.round(1 / stride))
                    conv = nn.Conv2d(planes, up, st, stride=st, bias=False)
                deblocks.append(nn.Sequential(conv, _bn(self._norm_cfg, up), nn.ReLU()))
        self.blocks = nn.ModuleList(blocks)
        self.deblocks = nn.ModuleList(deblocks)

    @property
    def downsample_factor(self):
        factor = np.prod(self._layer_strides)
        if len(self._upsample_strides) > 0:
            factor /= self._upsample_strides[-1]
        return factor

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight)

    # ------------------------------------------------------------------ reference composition (NCHW, torch ops)
    def forward_reference(self, x):
        ups = []
        for i in range(len(self.blocks)):
            x = self.blocks[i](x)
            if i - self._upsample_start_idx >= 0:
                ups.append(self.deblocks[i - self._upsample_start_idx](x))
        return torch.cat(ups, dim=1) if ups else x

    # ------------------------------------------------------------------ row kernels
    @staticmethod
    def _groups(seq):
        """[(conv, bn, relu, pad)] of a block / deblock Sequential."""
        mods = list(seq)
        out, pad, i = [], 0, 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.ZeroPad2d):
                pad = int(m.padding[0])
                i += 1
                continue
            assert isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)) and isinstance(mods[i + 1], nn.BatchNorm2d)
            relu = i + 2 < len(mods) and isinstance(mods[i + 2], nn.ReLU)
            out.append((m, mods[i + 1], relu, pad + int(m.padding[0])))
            pad = 0
            i += 3 if relu else 2
        return out

    def _plan(self):
        plan = self.__dict__.get("_row_plan")
        if plan is None:
            plan = {"blocks": [[(_Layer(c, b, r), p) for c, b, r, p in self._groups(blk)] for blk in self.blocks],
                    "deblocks": [[(_Layer(c, b, r), p) for c, b, r, p in self._groups(d)] for d in self.deblocks],
                    "nbr": {}}
            self.__dict__["_row_plan"] = plan
        return plan

    def train(self, mode=True):
        self.__dict__.pop("_row_plan", None)
        return super(RPN, self).train(mode)

    @staticmethod
    def _table(layer, pad, B, H, W, tables, device):
        layer.prepare(pad)
        key = (B, H, W, layer.kh, layer.kw, layer.stride, layer.pad, layer.transposed)
        if key not in tables:
            tables[key] = _ops.conv2d_neighbors(B, H, W, layer.kh, layer.kw, layer.stride, layer.pad, layer.transposed,
                                                device)
        return tables[key]

    def accepts_split_rows(self):
        """True when forward_rows can take its input as split rows (`SplitRows`): the first layer runs on the
        split-precision kernel, which reads nothing else."""
        if _ops.CONV_PRECISION != "split" or self.training or os.environ.get("DF3D_FUSION_TAIL", "1") != "1":
            return False
        plan = self._plan()
        if not plan["blocks"] or not plan["blocks"][0]:
            return False
        layer, pad = plan["blocks"][0][0]
        layer.prepare(pad)
        return layer.packed is not None and layer.packed16 is None

    @staticmethod
    def _cat_width(plan):
        """Channels of the concatenated upsampled maps if their last layers can write them in place (split-precision
        output-stationary kernels, 8-channel column offsets; round 3: no cat copy, no operand split in the head), else 0."""
        if plan.get("cat", (None, 0))[0] != _ops.CONV_PRECISION:       # the packs follow the precision switch
            lasts = [stack[-1] for stack in plan["deblocks"] if stack]
            for layer, pad in lasts:
                layer.prepare(pad)
            ok = (len(lasts) > 1 and len(lasts) == len(plan["deblocks"]) and os.environ.get("DF3D_NECK_CAT", "1") != "0"
                  and os.environ.get("DF3D_SPLIT_KERNEL", "o")[:1] != "p"
                  and all(l.packed is not None and l.packed16 is None and l.filters.shape[2] % 8 == 0 for l, _ in lasts))
            plan["cat"] = (_ops.CONV_PRECISION, sum(l.filters.shape[2] for l, _ in lasts) if ok else 0)
        return plan["cat"][1]

    @staticmethod
    def _run(layer, pad, rows, split, B, H, W, tables):
        nbr, Ho, Wo = RPN._table(layer, pad, B, H, W, tables, rows.device)
        K, cin, cout = layer.filters.shape
        n_out = nbr.shape[1]
        if layer.packed16 is not None:                       # DF3D_CONV_PRECISION=bf16: bf16 rows from layer to layer
            r16 = split if (split is not None and split.dtype == torch.bfloat16) else _ops.rows_to_bf16(rows)
            out, o16 = _ops.sparse_conv_bf16(r16, layer.packed16, nbr, n_out, cin, cout, bias=layer.bias,
                                             scale=layer.scale, shift=layer.shift, relu=layer.relu, want_f32=True)
            return out, o16, Ho, Wo
        if split is not None and split.dtype == torch.bfloat16:
            split = None
        if layer.packed is not None:
            if split is None:
                split = _ops.split_rows(rows)
            out, osplit = _ops.sparse_conv_split(split, layer.packed, nbr, n_out, cin, cout, bias=layer.bias,
                                                 scale=layer.scale, shift=layer.shift, relu=layer.relu)
        else:
            out = _ops.sparse_conv_fused(rows, layer.filters, nbr, n_out, bias=layer.bias, scale=layer.scale,
                                         shift=layer.shift, relu=layer.relu)
            osplit = None
        return out, osplit, Ho, Wo

    @torch.no_grad()
    def forward_rows(self, rows, B, H, W):
        """rows [B*H*W, C] fp32 channels-last (row = (b, y, x)) -> NCHW-shaped, channels-last-strided output."""
        plan = self._plan()
        tables = plan["nbr"]
        if isinstance(rows, SplitRows):              # the producer wrote the operand format of the first layer directly
            x, xs = _NoRows(rows.split.device), rows.split
        else:
            x, xs = rows.contiguous(), None
        ups = []
        cat, cat_rows, cat_split, col0 = self._cat_width(plan), None, None, 0
        for i, blk in enumerate(plan["blocks"]):
            for layer, pad in blk:
                x, xs, H, W = self._run(layer, pad, x, xs, B, H, W, tables)
            j = i - self._upsample_start_idx
            if j >= 0:
                u, us, uh, uw = x, xs, H, W
                stack = plan["deblocks"][j]
                for layer, pad in stack[:-1] if cat else stack:
                    u, us, uh, uw = self._run(layer, pad, u, us, B, uh, uw, tables)
                if cat:                          # the last layer writes its columns of the concatenated rows itself
                    layer, pad = stack[-1]
                    nbr, uh, uw = self._table(layer, pad, B, uh, uw, tables, x.device)
                    if cat_rows is None:
                        cat_rows = torch.empty((nbr.shape[1], cat), dtype=torch.float32, device=x.device)
                        cat_split = torch.empty((nbr.shape[1], _ops.split_width(cat)), dtype=torch.uint8, device=x.device)
                    K, cin, cout = layer.filters.shape
                    _ops.conv_rows_split(us if us is not None else _ops.split_rows(u), cin, 0, layer.packed, cout, 1, nbr,
                                         nbr.shape[1], layer.bias, layer.scale, layer.shift, layer.relu,
                                         into=(cat_rows, cat_split, col0))
                    col0 += cout
                    u = None
                ups.append((u, uh, uw))
        if not ups:
            return x.view(B, H, W, -1).permute(0, 3, 1, 2)
        uh, uw = ups[0][1], ups[0][2]
        assert all(h == uh and w == uw for _, h, w in ups)
        if cat:
            return _as_nchw(cat_rows, cat_split, B, uh, uw)
        out = torch.cat([u for u, _, _ in ups], 1) if len(ups) > 1 else ups[0][0]
        return out.view(B, uh, uw, -1).permute(0, 3, 1, 2)

    def forward_rows_train(self, rows, B, H, W):
        """`forward_rows` with autograd and BatchNorm in whatever mode the module is in (`_train_stack`)."""
        tables = self.__dict__.setdefault("_train_tables", {})
        ups = []
        for i, blk in enumerate(self.blocks):
            rows, H, W = _train_stack(self._groups(blk), rows, B, H, W, tables)
            j = i - self._upsample_start_idx
            if j >= 0:
                ups.append(_train_stack(self._groups(self.deblocks[j]), rows, B, H, W, tables))
        # contiguous NCHW out: a channels-last view would pull the head's library convolutions onto their NHWC kernels
        # (a filter re-layout per call and branch)
        if not ups:
            return rows.view(B, H, W, -1).permute(0, 3, 1, 2).contiguous()
        uh, uw = ups[0][1], ups[0][2]
        assert all(h == uh and w == uw for _, h, w in ups)
        out = torch.cat([u for u, _, _ in ups], 1) if len(ups) > 1 else ups[0][0]
        return out.view(B, uh, uw, -1).permute(0, 3, 1, 2).contiguous()

    def forward(self, x):
        if not x.is_cuda or x.dtype != torch.float32:
            return self.forward_reference(x)
        B, C, H, W = x.shape
        rows = x.permute(0, 2, 3, 1).reshape(B * H * W, C)
        if self.training or torch.is_grad_enabled():
            return self.forward_rows_train(rows, B, H, W)
        return self.forward_rows(rows, B, H, W)


# ---------------------------------------------------------------------------------------------- TransFusion tree
def _norm2d(norm_cfg, planes):
    cfg = dict(norm_cfg or dict(type="BN", eps=1e-3, momentum=0.01))
    t = cfg.pop("type", "BN")
    if t not in ("BN", "BN2d"):
        raise KeyError("unsupported norm type for the BEV neck: %s" % t)
    cfg.pop("requires_grad", None)
    return nn.BatchNorm2d(planes, **cfg)


def _conv2d(conv_cfg, cin, cout, k, **kw):
    cfg = dict(conv_cfg or dict(type="Conv2d", bias=False))
    t = cfg.pop("type", "Conv2d")
    if t not in ("Conv2d", "Conv"):
        raise KeyError("unsupported conv type for the BEV neck: %s" % t)
    kw.update(cfg)
    return nn.Conv2d(cin, cout, k, **kw)


def _rows_of(x):
    """NCHW tensor -> (rows [B*H*W, C], split rows or None); free for the channels-last views this module returns."""
    cached = getattr(x, "_df3d_rows", None)
    if cached is not None and cached[2] == x._version:       # views share the version counter: in-place edits void it
        return cached[0], cached[1]
    B, C, H, W = x.shape
    return x.permute(0, 2, 3, 1).reshape(B * H * W, C), None


def _as_nchw(rows, split, B, H, W):
    out = rows.view(B, H, W, -1).permute(0, 3, 1, 2)
    out._df3d_rows = (rows, split, out._version)   # lets the next module skip the layout change and the operand split
    return out


class _RowStacks(nn.Module):
    """Shared machinery: per-Sequential row plans, invalidated by train()."""

    def _stacks(self, name, seqs):
        plan = self.__dict__.get("_row_plan")
        if plan is None:
            plan = self.__dict__["_row_plan"] = {"nbr": {}}
        if name not in plan:
            plan[name] = [[(_Layer(c, b, r), p) for c, b, r, p in RPN._groups(s)] for s in seqs]
        return plan[name], plan["nbr"]

    def train(self, mode=True):
        self.__dict__.pop("_row_plan", None)
        return super(_RowStacks, self).train(mode)

    @staticmethod
    def _fast(x):
        return x.is_cuda and x.dtype == torch.float32 and not torch.is_grad_enabled()


@MM_BACKBONES.register_module()
class SECOND(_RowStacks):
    def __init__(self, in_channels=128, out_channels=[128, 128, 256], layer_nums=[3, 5, 5], layer_strides=[2, 2, 2],
                 norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), conv_cfg=dict(type="Conv2d", bias=False)):
        super(SECOND, self).__init__()
        assert len(layer_strides) == len(layer_nums) == len(out_channels)
        in_filters = [in_channels] + list(out_channels[:-1])
        blocks = []
        for i, layer_num in enumerate(layer_nums):
            mods = [_conv2d(conv_cfg, in_filters[i], out_channels[i], 3, stride=layer_strides[i], padding=1),
                    _norm2d(norm_cfg, out_channels[i]), nn.ReLU(inplace=True)]
            for _ in range(layer_num):
                mods += [_conv2d(conv_cfg, out_channels[i], out_channels[i], 3, padding=1),
                         _norm2d(norm_cfg, out_channels[i]), nn.ReLU(inplace=True)]
            blocks.append(nn.Sequential(*mods))
        self.blocks = nn.ModuleList(blocks)

    def init_weights(self, pretrained=None):
        if isinstance(pretrained, str):
            self.load_state_dict(torch.load(pretrained, map_location="cpu").get("state_dict", {}), strict=False)

    def forward_reference(self, x):
        outs = []
        for blk in self.blocks:
            x = blk(x)
            outs.append(x)
        return tuple(outs)

    def forward(self, x):
        if (self.training or torch.is_grad_enabled()) and _train_rows_ok(x):
            tables = self.__dict__.setdefault("_train_tables", {})
            B, _, H, W = x.shape
            rows, _ = _rows_of(x)
            outs = []
            for blk in self.blocks:
                rows, H, W = _train_stack(RPN._groups(blk), rows, B, H, W, tables)
                outs.append(rows.view(B, H, W, -1).permute(0, 3, 1, 2))
            return tuple(outs)
        if self.training or not self._fast(x):
            return self.forward_reference(x)
        stacks, tables = self._stacks("blocks", self.blocks)
        B, _, H, W = x.shape
        rows, split = _rows_of(x)
        rows = rows.contiguous()
        outs = []
        for stack in stacks:
            for layer, pad in stack:
                rows, split, H, W = RPN._run(layer, pad, rows, split, B, H, W, tables)
            outs.append(_as_nchw(rows, split, B, H, W))
        return tuple(outs)


@MM_NECKS.register_module()
class SECONDFPN(_RowStacks):
    def __init__(self, in_channels=[128, 128, 256], out_channels=[256, 256, 256], upsample_strides=[1, 2, 4],
                 norm_cfg=dict(type="BN", eps=1e-3, momentum=0.01), upsample_cfg=dict(type="deconv", bias=False),
                 conv_cfg=dict(type="Conv2d", bias=False), use_conv_for_no_stride=False):
        super(SECONDFPN, self).__init__()
        assert len(out_channels) == len(upsample_strides) == len(in_channels)
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.fp16_enabled = False
        deblocks = []
        for i, out_channel in enumerate(out_channels):
            stride = upsample_strides[i]
            if stride > 1 or (stride == 1 and not use_conv_for_no_stride):
                cfg = dict(upsample_cfg)
                if cfg.pop("type", "deconv") != "deconv":
                    raise KeyError("the BEV neck serves upsample_cfg type 'deconv'")
                up = nn.ConvTranspose2d(in_channels[i], out_channel, int(stride), stride=int(stride), **cfg)
            else:
                st = int(np.round(1 / stride))
                up = _conv2d(conv_cfg, in_channels[i], out_channel, st, stride=st)
            deblocks.append(nn.Sequential(up, _norm2d(norm_cfg, out_channel), nn.ReLU(inplace=True)))
        self.deblocks = nn.ModuleList(deblocks)

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward_reference(self, x):
        assert len(x) == len(self.in_channels)
        ups = [d(x[i]) for i, d in enumerate(self.deblocks)]
        return [torch.cat(ups, dim=1) if len(ups) > 1 else ups[0]]

    def forward(self, x):
        assert len(x) == len(self.in_channels)
        if (self.training or torch.is_grad_enabled()) and all(_train_rows_ok(t) for t in x):
            tables = self.__dict__.setdefault("_train_tables", {})
            ups = []
            for t, d in zip(x, self.deblocks):
                B, _, H, W = t.shape
                ups.append(_train_stack(RPN._groups(d), _rows_of(t)[0], B, H, W, tables))
            H, W = ups[0][1], ups[0][2]
            assert all(h == H and w == W for _, h, w in ups)
            rows = torch.cat([u for u, _, _ in ups], 1) if len(ups) > 1 else ups[0][0]
            return [rows.view(B, H, W, -1).permute(0, 3, 1, 2)]
        if self.training or not all(self._fast(t) for t in x):
            return self.forward_reference(x)
        stacks, tables = self._stacks("deblocks", self.deblocks)
        ups = []
        for t, stack in zip(x, stacks):
            B, _, H, W = t.shape
            rows, split = _rows_of(t)
            rows = rows.contiguous()
            for layer, pad in stack:
                rows, split, H, W = RPN._run(layer, pad, rows, split, B, H, W, tables)
            ups.append((rows, H, W))
        H, W = ups[0][1], ups[0][2]
        assert all(h == H and w == W for _, h, w in ups)
        rows = torch.cat([u for u, _, _ in ups], 1) if len(ups) > 1 else ups[0][0]
        return [_as_nchw(rows, None, B, H, W)]
