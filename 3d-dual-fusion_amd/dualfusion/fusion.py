"""CenterPoint camera->LiDAR fusion adapter on the MI355X kernels.

Registry name, constructor keys, forward signature and parameter names follow the reference
(`VoxelWithPointProjection`, CP/det3d/models/fusion/voxel_with_point_projection.py:13-385; the
image-side gate `Basicgate_patch_iv_multivoxel`, CP/det3d/models/model_utils/attention.py:8-61;
projection CP/det3d/models/fusion/point_to_image_projection.py:63-231), so the reference config
`nusc_centerpoint_voxelnet_0075voxel_fix_bn_z_multimodal_pfat_hybrid7_ifat.py:71-109` builds it
unchanged and `fusion.pfat.* / fusion.ifat.*` checkpoints load.

Execution differs: the reference loops cameras x samples x scales in Python (boolean indexing,
`.unique()`, `.cpu()`); here each stage is one batched kernel over all B*6 images
(csrc/fusion.hip), the gate's 1x1 / 3x3 convolutions run once over the [B*6, .] image batch, and
the only host sync is the read of max_ne (the padded query length), which the reference needs too.
Semantics kept bug-for-bug.
"""
import contextlib
import ctypes

import os

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import _lib, synth
from . import ops as _ops
from .actr import build as build_actr
from .registry import FUSION


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


class _TallMatmul(torch.autograd.Function):
    """a [rows, k] @ b [k, n] for a few columns and ~10^5 rows.  The gradient of b contracts over the rows: as one library product
    ([2, 240300] x [240300, 9]: 0.56 ms) or as autograd's column reductions (0.2 ms each) it costs more than the layer it belongs
    to; here per chunk of rows (a batched product) and a sum over the chunks."""

    @staticmethod
    def forward(ctx, a, b, chunks):
        ctx.save_for_backward(a, b)
        ctx.chunks = chunks if a.shape[0] % max(chunks, 1) == 0 else 1
        return a @ b

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = gb = None
        if ctx.needs_input_grad[0]:
            ga = g @ b.t()
        if ctx.needs_input_grad[1]:
            P = ctx.chunks
            gb = torch.bmm(a.view(P, -1, a.shape[1]).transpose(1, 2), g.reshape(P, -1, g.shape[1])).sum(0)
        return ga, gb, None


class Basicgate_patch_iv_multivoxel(nn.Module):
    """Image-side gate (attention.py:8-61): voxel features of the scales in `voxel_idx` are dropped
    on the image plane, mixed by 1x1 convs, added to a 1-channel image summary and turned into a
    sigmoid attention map by a 3x3 conv; the image feature is multiplied by that map."""

    def __init__(self, **kwarg):
        super().__init__()
        self.img_num_channel = kwarg['img_num_channel']
        self.pts_num_channel = kwarg['pts_num_channel'] + 3
        self.voxel_feat_channel = kwarg['voxel_feat_channel']
        self.voxel_idx = kwarg['voxel_idx']
        if len(self.voxel_idx) == 1:
            self.pts_num_channel = self.voxel_feat_channel[self.voxel_idx[0]]
        last = self.voxel_feat_channel[self.voxel_idx[-1]] + 3
        self.reduced_dim2 = nn.Conv2d(last, last, kernel_size=1, stride=1, padding=0)
        self.reduced_dim3 = nn.Conv2d(self.img_num_channel, 1, kernel_size=1, stride=1, padding=0)
        self.spatial_basic = nn.Conv2d(last, 1, kernel_size=3, stride=1, padding=1)
        self.reduced_dim = nn.Sequential(*[
            nn.Conv2d(self.voxel_feat_channel[i] + 3, last, kernel_size=1, stride=1, padding=0)
            for i in range(self.voxel_idx[-1])])

    def folded(self):
        """The gate is linear up to its sigmoid, so it collapses to a few small matrices (see
        csrc/fusion.hip "Image-side gate without dense canvases"):
          T[idx] [9, C_idx+3]: tap responses of a voxel row of scale idx;  kg [19] = k_t, g_t, bias;
          w3 [1, Cimg], b3: the 1-channel image summary (reduced_dim3)."""
        ps = self.__dict__.get("_param_list")
        if ps is None:                                # collected once: the module-tree walk of parameters() is host time per step
            ps = self.__dict__["_param_list"] = list(self.parameters())
        key = tuple((p.data_ptr(), p._version) for p in ps)
        hit = getattr(self, "_folded", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        out = self._fold()
        object.__setattr__(self, "_folded", (key, out))
        return out

    def _fold(self):
        with torch.no_grad():
            last = self.voxel_idx[-1]
            R2 = self.reduced_dim2.weight[:, :, 0, 0].float()
            Wsb = self.spatial_basic.weight[0].float().permute(1, 2, 0).reshape(9, -1)     # [tap, channel]
            const = self.reduced_dim2.bias.float().clone()
            T = {}
            for idx in self.voxel_idx:
                if len(self.voxel_idx) > 1 and idx != last:
                    Ri = self.reduced_dim[idx].weight[:, :, 0, 0].float()
                    T[idx] = (Wsb @ (R2 @ Ri)).contiguous()
                    const = const + R2 @ self.reduced_dim[idx].bias.float()
                else:
                    T[idx] = (Wsb @ R2).contiguous()
            kg = torch.cat([Wsb @ const, Wsb.sum(1), self.spatial_basic.bias.float().view(1)]).contiguous()
            return T, kg, self.reduced_dim3.weight[:, :, 0, 0].float(), self.reduced_dim3.bias.float()

    def forward_batched(self, img_feat, canvases):
        """img_feat [NI, Cimg, H, W]; canvases {scale idx: [NI, C_s+3, H, W]} -> gated img_feat."""
        pt_img = None
        for conv_idx in self.voxel_idx:
            f = canvases[conv_idx]
            if len(self.voxel_idx) > 1 and conv_idx != self.voxel_idx[-1]:
                f = self.reduced_dim[conv_idx](f)
            pt_img = f if pt_img is None else pt_img + f
        pt_img = self.reduced_dim2(pt_img)
        fused = self.reduced_dim3(img_feat) + pt_img          # [NI,1,H,W] broadcast over the channels
        return img_feat * torch.sigmoid(self.spatial_basic(fused))


    def forward_rows(self, img_feat, canvases):
        """forward_batched on pixel-major canvases {scale idx: [NI, H*W, C_s+3]}, with every convolution written as a
        matrix product (the 1x1 convs) or as nine shifted sums of one (the 3x3 conv to one channel): same values,
        differentiable, and none of it goes through MIOpen, whose fp32 fallback kernels for these shapes take
        20-160 ms per call forward / backward."""
        return img_feat * self.attention_rows(img_feat, canvases).unsqueeze(1)

    def attention_sparse(self, img_feat, rows, winners):
        """attention_rows WITHOUT the dense canvases (training, round 6).  The gate is linear up to its sigmoid (see `folded`):
        the tap responses of (point term + image summary) at a pixel are
          taps[pixel, t] = sum_idx T_idx[t] . (winner row of scale idx at the pixel)  +  g[t] . summary(pixel)  +  k[t],
        T_idx = Wsb (R2 R_idx) [9, C_idx + 3], g = row sums of Wsb, k = Wsb . (folded biases), and the 3 x 3 convolution is the
        sum of the nine shifted tap maps (zero padding: as attention_rows).  So per scale ONE product [n_idx, C_idx + 3] x
        [C_idx + 3, 9] on the voxel rows, a gather of the winners and a write of their nine responses at their pixels (a pixel
        has one winner per scale: no accumulation, the same bits in every run) replace the [NI, H W, 131] canvases, four
        products over 240 k pixel rows and their backward.
        rows: {idx: [n_idx, C_idx + 3] voxel rows + inverse points}; winners: {idx: (image, pixel, row) of every occupied pixel}.
        Same function of the parameters and rows as attention_rows (summation order aside); differentiable in both."""
        NI, Ci, H, W = img_feat.shape
        last = self.voxel_idx[-1]
        R2 = self.reduced_dim2.weight[:, :, 0, 0]
        Wsb = self.spatial_basic.weight[0].permute(1, 2, 0).reshape(9, -1)                      # [tap = ty * 3 + tx, channel]
        const = self.reduced_dim2.bias
        summary = _ops.channel_first_linear(img_feat.reshape(NI, Ci, H * W), self.reduced_dim3.weight[:, :, 0, 0])   # [NI, 1, HW]
        taps = None
        for idx in self.voxel_idx:
            if len(self.voxel_idx) > 1 and idx != last:
                cv = self.reduced_dim[idx]
                T = Wsb @ (R2 @ cv.weight[:, :, 0, 0])
                const = const + R2 @ cv.bias
            else:
                T = Wsb @ R2
            img_w, pix_w, row_w = winners[idx]
            resp = rows[idx] @ T.t()                                                             # [n_idx, 9]
            t_idx = resp.new_zeros((NI * H * W, 9)).index_put((img_w * (H * W) + pix_w,), resp[row_w])
            taps = t_idx if taps is None else taps + t_idx
        # g[t] . summary(pixel) + k[t] as ONE product [NI H W, 2] x [2, 9]: its backward w.r.t. g and k is then a product over the
        # 240 k pixel rows too (autograd's broadcast form reduced three [240300, 9] tensors column-wise: 0.2 ms each)
        s1 = torch.cat([summary.reshape(-1, 1) + self.reduced_dim3.bias, summary.new_ones((NI * H * W, 1))], 1)
        dense = _TallMatmul.apply(s1, torch.stack([Wsb.sum(1), Wsb @ const], 0), NI * H)        # [NI H W, 9]
        taps = F.pad((taps + dense).view(NI, H, W, 9), (0, 0, 1, 1, 1, 1))                       # zero padding of the 3x3
        y = self.spatial_basic.bias.view(1, 1, 1)
        for ty in range(3):
            for tx in range(3):
                y = y + taps[:, ty:ty + H, tx:tx + W, ty * 3 + tx]
        return torch.sigmoid(y)

    def attention_rows(self, img_feat, canvases):
        """The gate itself, [NI, H, W] (forward_rows without the product with the image)."""
        NI, Ci, H, W = img_feat.shape
        pt = None
        for conv_idx in self.voxel_idx:
            f = canvases[conv_idx]
            if len(self.voxel_idx) > 1 and conv_idx != self.voxel_idx[-1]:
                cv = self.reduced_dim[conv_idx]
                f = _linear_rows(f, cv.weight[:, :, 0, 0], cv.bias)
            pt = f if pt is None else pt + f
        pt = _linear_rows(pt, self.reduced_dim2.weight[:, :, 0, 0], self.reduced_dim2.bias)          # [NI, HW, last]
        summary = _ops.channel_first_linear(img_feat.reshape(NI, Ci, H * W), self.reduced_dim3.weight[:, :, 0, 0])   # [NI, 1, HW]
        fused = pt + (summary.transpose(1, 2) + self.reduced_dim3.bias)
        taps = _linear_rows(fused, self.spatial_basic.weight[0].permute(1, 2, 0).reshape(9, -1))     # [NI, HW, 9]
        taps = F.pad(taps.view(NI, H, W, 9), (0, 0, 1, 1, 1, 1))                                     # zero padding of the 3x3
        y = self.spatial_basic.bias.view(1, 1, 1)
        for ty in range(3):
            for tx in range(3):
                y = y + taps[:, ty:ty + H, tx:tx + W, ty * 3 + tx]
        return torch.sigmoid(y)


def _stack_views(maps):
    """torch.stack(maps, 0) -- without the copy when the maps are equally spaced contiguous views of ONE storage (the camera
    network runs the cameras as one batch: the per-camera dict entries are slices of its output; 246 MB per step otherwise)."""
    m0 = maps[0]
    if len(maps) > 1 and m0.is_contiguous():
        step = (maps[1].data_ptr() - m0.data_ptr()) // m0.element_size()
        if (step >= m0.numel() and all(m.is_contiguous() and m.shape == m0.shape and m.dtype == m0.dtype and m.device == m0.device
                                       and m.untyped_storage().data_ptr() == m0.untyped_storage().data_ptr()
                                       and (m.data_ptr() - m0.data_ptr()) // m0.element_size() == i * step
                                       for i, m in enumerate(maps))
                and not any(m.requires_grad for m in maps)):
            return torch.as_strided(m0, (len(maps),) + tuple(m0.shape), (step,) + tuple(m0.stride()))
    return torch.stack(maps, 0)


# (the row-linear autograd Function and its two-stage column sum live in ops.py: the ACTR modules use them too)
_col_sum = _ops.col_sum_rows
_linear_rows = _ops.linear_rows_autograd


ifat_all = {'Basicgate_patch_iv_multivoxel': Basicgate_patch_iv_multivoxel}


@FUSION.register_module
class VoxelWithPointProjection(nn.Module):
    def __init__(self, fuse_mode, interpolate, voxel_size, pc_range, image_list, image_scale=1, depth_thres=0,
                 double_flip=False, layer_channel=None, pfat_cfg=None, lt_cfg=None, ifat_cfg=None, seg_cfg=None,
                 model_name='ACTR'):
        super().__init__()
        if fuse_mode != 'pfat' or interpolate or double_flip or seg_cfg:
            raise NotImplementedError("only fuse_mode='pfat', interpolate=False, no double_flip/seg head "
                                      "(the 3D-DF configuration) is implemented on the MI355X path")
        self.voxel_size = [float(v) for v in voxel_size]
        self.pc_range = [float(v) for v in pc_range]
        self.fuse_mode = fuse_mode
        self.image_interp = interpolate
        self.image_list = list(image_list)
        self.image_scale = image_scale
        self.double_flip = double_flip
        self.depth_thres = depth_thres
        self.pfat = build_actr(pfat_cfg, lt_cfg=lt_cfg, model_name=model_name)
        self.ifat_cfg = None
        if ifat_cfg:
            self.ifat_cfg = ifat_cfg
            self.ifat = ifat_all[ifat_cfg['fusion_method']](**ifat_cfg)
        self._calib_cache = None
        self._shape_cache = None
        self._side = None
        # set by a caller whose camera maps / calibration tensors are complete in device memory before forward() (bench.py):
        # the small per-frame tensors derived from them are then produced on the adapter's side stream, which later reads them
        # without having to wait for the caller's stream (needed once the backbone's geometry no longer waits for it either)
        self.resident_inputs = False
        self._prefetched = {}                 # (id(batch_dict), layer name) -> (inputs, image projection, event or None)
        self._img_stream = None
        self._prepared = None
        self._ptr_tables = {}
        self._wcat = None
        self._wpack = None

    # ------------------------------------------------------------------ inputs
    def _gather_inputs(self, batch_dict, layer_name, dev):
        """Per-camera dict entries -> what the kernels take (image index = b*ncam + cam).  The camera feature
        maps stay where they are: `imgs` lists the B*ncam [Ci, h, w] views and `img_ptrs` is their device
        pointer table (no stacking copy).  The small calibration tensors are cached per calib dict."""
        cams = [c.lower() for c in self.image_list]
        feats = batch_dict['img_feat'][layer_name + '_feat2d']
        B = feats[cams[0]].shape[0]
        ncam = len(cams)
        imgs = []
        for b in range(B):
            for c in cams:
                f = feats[c][b]
                if f.dtype != torch.float32 or not f.is_contiguous():
                    f = f.float().contiguous()
                imgs.append(f)
        Ci, h, w = imgs[0].shape
        img_ptrs = self._pointer_table(imgs, dev)
        # Calibration differs per sample (nuScenes lidar2cam changes every frame), so nothing here is keyed on object
        # identity: the image-shape derived tensors are cached by VALUE (the shapes are a handful of host integers),
        # the calibration matrices are re-stacked unless the very same tensors (held by reference, same version
        # counters) come again.
        calib = batch_dict['calib']
        shp_cpu = torch.stack([torch.as_tensor(batch_dict['image_shape'][c])[:, :2] for c in cams], 1).cpu()  # [B,ncam,2]
        skey = (tuple(shp_cpu.reshape(-1).tolist()), h, w, str(dev))
        if self._shape_cache is None or self._shape_cache[0] != skey:
            shp_np = shp_cpu.numpy()
            fs = np.empty((B, ncam, 2), np.float32)
            for b in range(B):
                for c in range(ncam):
                    fs[b, c, 0] = np.float32(w / float(shp_np[b, c, 1]))          # :265-266 (python float -> fp32)
                    fs[b, c, 1] = np.float32(h / float(shp_np[b, c, 0]))
            thres = torch.tensor([float(self.depth_thres[c.upper()]) if isinstance(self.depth_thres, dict)
                                  else float(self.depth_thres) for c in cams], dtype=torch.float32, device=dev)
            self._shape_cache = (skey, dict(raw_hw=shp_cpu.to(torch.int32).contiguous().to(dev),
                                            feat_scale=torch.from_numpy(fs).to(dev), thres=thres))
        mats = [calib['lidar2cam_' + c.lstrip('cam_')] for c in cams] + \
               [calib['cam_intrinsic_' + c.lstrip('cam_')] for c in cams]
        hit = self._calib_cache
        if hit is None or len(hit[0]) != len(mats) or any(a is not b or a._version != v
                                                          for a, b, v in zip(mats, hit[0], hit[1])):
            if self.resident_inputs and dev.type == "cuda":
                if self._side is None:
                    self._side = torch.cuda.Stream(device=dev)
                main = torch.cuda.current_stream(dev)
                with torch.cuda.stream(self._side):
                    l2c = torch.stack([m.float() for m in mats[:ncam]], 1).contiguous().to(dev)
                    intr = torch.stack([m.float() for m in mats[ncam:]], 1).contiguous().to(dev)
                main.wait_stream(self._side)
                l2c.record_stream(main)
                intr.record_stream(main)
            else:
                l2c = torch.stack([m.float() for m in mats[:ncam]], 1).contiguous().to(dev)
                intr = torch.stack([m.float() for m in mats[ncam:]], 1).contiguous().to(dev)
            self._calib_cache = hit = (mats, [m._version for m in mats], dict(l2c=l2c, intr=intr))
        self._calib_cache = (hit[0], hit[1], dict(hit[2], **self._shape_cache[1]))
        out = dict(self._calib_cache[2])
        out.update(imgs=imgs, img_ptrs=img_ptrs, B=B, ncam=ncam, Ci=Ci, h=h, w=w)
        if 'aug_matrix_inv' in batch_dict:
            out['aug_inv'] = self._aug_inverse(batch_dict['aug_matrix_inv'], B, dev)
        return out

    @staticmethod
    def _aug_inverse(aug_matrix_inv, B, dev):
        """batch_dict['aug_matrix_inv'] (one dict per sample with any of 'translate' [3] / 'rescale' / 'rotate' / 'flip'
        [3,3], as CP/det3d/datasets/pipelines/preprocess.py:309-354 records them) -> [B, 30] fp32 for
        df3d_project_voxels: translate, then the three row-vector factors in the order the reference applies them
        (point_to_image_projection.py:121-128)."""
        rows = np.zeros((B, 30), np.float32)
        for b in range(B):
            rec = aug_matrix_inv[b] if not isinstance(aug_matrix_inv, dict) else aug_matrix_inv
            if 'translate' in rec:
                rows[b, :3] = np.asarray(rec['translate'], np.float32).reshape(-1)[:3]
            for f, key in enumerate(('rescale', 'rotate', 'flip')):
                m = np.asarray(rec[key], np.float32).reshape(3, 3) if key in rec else np.eye(3, dtype=np.float32)
                rows[b, 3 + 9 * f:12 + 9 * f] = m.reshape(-1)
        return _ops.device_constant(rows, torch.float32, dev)          # cached by value: no per-frame pageable copy

    def _pointer_table(self, imgs, dev):
        """Device table of the maps' addresses.  Feature buffers are normally recycled by the caching allocator,
        so the table of a previous frame is reused when the addresses repeat (an H2D copy from pageable memory
        would otherwise stall the host until the stream drains)."""
        ptrs = tuple(f.data_ptr() for f in imgs)
        hit = self._ptr_tables.get(ptrs)
        if hit is None:
            if len(self._ptr_tables) >= 8:
                self._ptr_tables.clear()
            hit = torch.tensor(ptrs, dtype=torch.int64, device=dev)
            self._ptr_tables[ptrs] = hit
        return hit

    # ------------------------------------------------------------------ image-side projection
    def _image_projection(self, inp, img_conv_func=None, pixrow=None):
        """`both` [NI, C(+1+pad), H*W] = input_proj 1x1 conv of every camera map WITHOUT bias (+ the gate's 1-channel
        image summary as an extra row when the image gate is configured): one GEMM per map, written in place."""
        imgs = inp['imgs']
        if img_conv_func is not None:
            imgs = list(img_conv_func(torch.stack(imgs, 0)))
            inp['imgs'] = imgs
            inp['img_ptrs'] = self._pointer_table(imgs, imgs[0].device)
        w_full = self.pfat.input_proj[0][0].weight
        w_ip = w_full[:, :, 0, 0]
        if self.ifat_cfg is not None:
            w3 = self.ifat.folded()[2]
            key = (w_full.data_ptr(), w_full._version, w3.data_ptr())
            if self._wcat is None or self._wcat[0] != key:
                # rows padded to a multiple of 16: hipBLASLt picks a 2x slower macro-tile for a 129-row operand
                npad = (-(w_ip.shape[0] + 1)) % 16
                self._wcat = (key, torch.cat([w_ip, w3, w3.new_zeros((npad, w3.shape[1]))], 0))
            wcat = self._wcat[1]
        else:
            wcat = w_ip
        S_pix = inp['h'] * inp['w']
        if (img_conv_func is None and self.pfat.can_fold() and len(self.pfat.transformer.encoder.layers) == 2
                and _ops.imgproj_supported(wcat.shape[0], imgs[0].shape[0], w_ip.shape[0])
                and os.environ.get("DF3D_IMGPROJ", "1") == "1"):
            # native split-precision projection straight from the channel-first maps (csrc/imgproj.hip): returns
            # (pixel-major split rows of the 128 projection channels, fp32 gate row)
            key = (wcat.data_ptr(), wcat._version)
            if self._wpack is None or self._wpack[0] != key:
                self._wpack = (key, _ops.imgproj_pack(wcat.contiguous()))
            if pixrow is not None and os.environ.get("DF3D_ASSEMBLE_COMPACT", "1") == "1":
                # (round 4) also the raw rows of the pixels the queries sample, pixel-major, for the query assembly.  With the
                # LDS-staged projection and the wave-per-candidate assembly this was a loss (assembly 77 -> 63 us, projection
                # 121 -> 155 us); with the direct projection (the lane holds its pixel's channels: two stores under a mark,
                # 92 -> 103 us) and the by-slot assembly (78 -> 42 us) it is a gain of ~50 us per step: on by default,
                # DF3D_ASSEMBLE_COMPACT=0 turns it off
                return _ops.imgproj_split(inp['img_ptrs'], len(imgs), imgs[0].shape[0], S_pix, self._wpack[1],
                                          pixrow=pixrow[0], pixrow_total=pixrow[1])
            return _ops.imgproj_split(inp['img_ptrs'], len(imgs), imgs[0].shape[0], S_pix, self._wpack[1])
        # The camera network runs the cameras as one batch, so the per-camera dict entries are normally views of one
        # tensor at a uniform stride: then the projection is a single batched GEMM over that storage (still no copy).
        # Unrelated buffers get one GEMM per map.
        Ci = imgs[0].shape[0]
        p0 = imgs[0].data_ptr()
        step = (imgs[1].data_ptr() - p0) if len(imgs) > 1 else Ci * S_pix * 4
        store = imgs[0].untyped_storage().data_ptr()
        if step >= Ci * S_pix * 4 and step % 4 == 0 and all(
                f.untyped_storage().data_ptr() == store and f.data_ptr() == p0 + i * step
                for i, f in enumerate(imgs)):
            stacked = torch.as_strided(imgs[0], (len(imgs), Ci, S_pix), (step // 4, S_pix, 1))
            return torch.bmm(wcat.unsqueeze(0).expand(stacked.shape[0], -1, -1), stacked)     # (matmul would transpose-copy the maps)
        both = torch.empty((len(imgs), wcat.shape[0], S_pix), dtype=torch.float32, device=imgs[0].device)
        for i, f in enumerate(imgs):
            torch.matmul(wcat, f.view(Ci, S_pix), out=both[i])
        return both

    def prefetch(self, batch_dict, layer_name='layer1_ori', img_conv_func=None, inp=None, ahead=False):
        """Start the image-side projection (it depends on the camera maps only) on a side stream, so that it
        overlaps the LiDAR branch; forward() picks the result up.  Optional: forward() computes it itself
        when prefetch was not called for this batch_dict.
        ahead: the camera maps are COMPLETE in device memory (resident inputs) and this is a LATER frame's batch_dict: the
        side stream does not wait for the caller's stream, so the projection runs beside whatever the GPU is doing now --
        with a frame of lookahead that is the previous frame's low-channel sparse convolutions, which leave most of the chip
        idle (round 4; inside ONE frame the same co-run loses)."""
        feats = batch_dict['img_feat'][layer_name + '_feat2d']
        dev = next(iter(feats.values())).device
        if ahead:
            if self._img_stream is None:
                self._img_stream = torch.cuda.Stream(device=dev)
            side = self._img_stream
        else:
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            side = self._side
            side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            if inp is None:
                inp = self._gather_inputs(batch_dict, layer_name, dev)
            elif ahead and inp.get('_ready') is not None:
                side.wait_event(inp['_ready'])            # calibration-independent, but the pointer table / maps of `inp`
            both = self._image_projection(inp, img_conv_func)
            ev = torch.cuda.Event()
            ev.record(side)
        self._remember_prefetched(batch_dict, layer_name, inp, both, ev)

    def _remember_prefetched(self, batch_dict, layer_name, inp, both, ev):
        if len(self._prefetched) >= 4:                     # entries nobody consumed (a caller that changed its frame order)
            self._prefetched.pop(next(iter(self._prefetched)))
        # the entry holds the dict itself: id() of a collected dict can be handed to the next one
        self._prefetched[(id(batch_dict), layer_name)] = (inp, both, ev, batch_dict)

    def prefetch_inline(self, batch_dict, layer_name='layer1_ori', img_conv_func=None):
        """The image-side projection issued early on the CURRENT stream (no co-running, just earlier in the frame)."""
        feats = batch_dict['img_feat'][layer_name + '_feat2d']
        dev = next(iter(feats.values())).device
        prep = self._prepared
        if prep is not None and prep[0] == id(batch_dict) and prep[1] == layer_name:
            inp = prep[2]['inp']                 # gathered by prepare_geometry (another host thread, a frame ahead)
        else:
            inp = self._gather_inputs(batch_dict, layer_name, dev)
        if (id(batch_dict), layer_name) in self._prefetched or (prep is not None and prep[0] == id(batch_dict)
                                                                 and "both" in prep[2]):
            return                                            # already projected (side stream / frame-head worker)
        pixrow = prep[2].get("pixrow") if (prep is not None and prep[0] == id(batch_dict) and img_conv_func is None) else None
        both = self._image_projection(inp, img_conv_func, pixrow=pixrow)
        self._remember_prefetched(batch_dict, layer_name, inp, both, None)

    def prepare_geometry(self, batch_dict, layer_name, levels, d_factor_list):
        """Everything of forward() that depends on the voxel COORDINATES and the calibration alone, on the current stream: the
        kernel inputs (`_gather_inputs`), the camera projection of every scale the gate or the queries read, the per-camera
        query slots and the adapter's one host round trip (longest query list).  `levels`: the encoded_voxel_list as tensors
        whose `.indices` are complete on the current stream (features are not read).  A caller that builds a frame's geometry
        ahead (dualfusion/prefetch.py) runs this there and hands the result over with `use_prepared`."""
        last = len(levels) - 1
        dev = levels[last].indices.device
        inp = self._gather_inputs(batch_dict, layer_name, dev)
        need = set([last])
        if self.ifat_cfg is not None:
            need |= set(self.ifat.voxel_idx)
        proj = {s_: self._project(levels[s_], d_factor_list[s_], inp) for s_ in sorted(need)}
        early = self._query_slots(levels[last].indices.contiguous(), proj[last][1], inp['B'])
        return dict(inp=inp, proj=proj, early=early)

    def head_request(self, batch_dict, layer_name, layers, d_factor_list, dev):
        """What `prepare_geometry` computes, as a job description for the native frame-head worker (dualfusion/prefetch.py):
        the kernel inputs gathered now (on the calling thread), the stages to project (`layers`: executor layer index of every
        entry of the encoded_voxel_list) with their voxel sizes, the stage whose visible voxels become queries."""
        inp = self._gather_inputs(batch_dict, layer_name, dev)
        last = len(layers) - 1
        need = set([last])
        if self.ifat_cfg is not None:
            need |= set(self.ifat.voxel_idx)
        levels = [(s_, layers[s_], [float(np.float32(np.float32(v) * np.float32(d_factor_list[s_]))) for v in self.voxel_size])
                  for s_ in sorted(need)]
        # the calibration stacks may have just been queued on the side stream: the worker's stream waits for them
        if self.resident_inputs and self._side is not None:
            ready = self._side.record_event()
        else:                                   # stacked on the caller's stream: the worker then waits for that stream
            ready = torch.cuda.current_stream(dev).record_event()
        req = dict(inp=inp, levels=levels, slots_level=last, pc_min=[float(np.float32(v)) for v in self.pc_range[:3]],
                   image_scale=self.image_scale, ready=ready,
                   # the image gate's "winning voxel per pixel" maps depend on the coordinates alone as well
                   winner_levels=tuple(self.ifat.voxel_idx) if self.ifat_cfg is not None else (),
                   want_pixrow=os.environ.get("DF3D_ASSEMBLE_COMPACT", "1") == "1")
        packed = self._native_projection_weights(inp)
        if packed is not None and os.environ.get("DF3D_IMGPROJ_AHEAD", "0") == "1":
            # the image-side projection depends on the camera maps alone: the worker can run it too, on a second stream of
            # default priority (opt-in: measured neutral on MI355X, 2.886 against 2.881 ms per step -- the projection then
            # co-runs with the previous frame's kernels and slows them by what it saves)
            req["image_projection"] = dict(packed=packed, cin=inp['Ci'], pixels=inp['h'] * inp['w'])
        return req

    def _native_projection_weights(self, inp):
        """Packed `Wcat` of `_image_projection`'s native kernel when that kernel serves this configuration, else None."""
        if not (self.pfat.can_fold() and len(self.pfat.transformer.encoder.layers) == 2
                and os.environ.get("DF3D_IMGPROJ", "1") == "1"):
            return None
        w_full = self.pfat.input_proj[0][0].weight
        w_ip = w_full[:, :, 0, 0]
        if self.ifat_cfg is not None:
            w3 = self.ifat.folded()[2]
            key = (w_full.data_ptr(), w_full._version, w3.data_ptr())
            if self._wcat is None or self._wcat[0] != key:
                npad = (-(w_ip.shape[0] + 1)) % 16
                self._wcat = (key, torch.cat([w_ip, w3, w3.new_zeros((npad, w3.shape[1]))], 0))
            wcat = self._wcat[1]
        else:
            wcat = w_ip
        if not _ops.imgproj_supported(wcat.shape[0], inp['imgs'][0].shape[0], w_ip.shape[0]):
            return None
        key = (wcat.data_ptr(), wcat._version)
        if self._wpack is None or self._wpack[0] != key:
            self._wpack = (key, _ops.imgproj_pack(wcat.contiguous()))
            torch.cuda.current_stream(wcat.device).synchronize()      # first use only: the worker's stream reads it
        return self._wpack[1]

    def use_prepared(self, batch_dict, layer_name, prepared):
        """Hand over a `prepare_geometry` result for the forward() of this batch_dict; its tensors must be complete on (and
        known to the allocator for) the stream forward() runs on."""
        self._prepared = (id(batch_dict), layer_name, prepared)

    @staticmethod
    def prepared_tensors(prepared):
        out = []
        for v in prepared['proj'].values():
            out.extend(v)
        out.extend([prepared['early'][0], prepared['early'][2]])
        inp = prepared['inp']
        out.extend(t for t in (inp.get('l2c'), inp.get('intr'), inp.get('raw_hw'), inp.get('feat_scale'), inp.get('thres'),
                               inp.get('img_ptrs'), inp.get('aug_inv')) if torch.is_tensor(t))
        return out

    def _project(self, x, d_factor, inp):
        lib = _lib.load()
        ind = x.indices.contiguous()
        n = ind.shape[0]
        dev = ind.device
        ncam = inp['ncam']
        # fp32 voxel size * d_factor (point_to_image_projection.py:82)
        scale = [float(np.float32(np.float32(v) * np.float32(d_factor))) for v in self.voxel_size]
        pmin = [float(np.float32(v)) for v in self.pc_range[:3]]
        grid = torch.empty((ncam, n, 2), dtype=torch.int32, device=dev)
        mask = torch.empty((ncam, n), dtype=torch.uint8, device=dev)
        pinv = torch.empty((n, 3), dtype=torch.float32, device=dev)
        sp, k1 = _lib.float_arr(scale)
        mp, k2 = _lib.float_arr(pmin)
        rc = lib.df3d_project_voxels(_p(ind), n, inp['B'], ncam, sp, mp, _p(inp['l2c']), _p(inp['intr']),
                                     _p(inp['raw_hw']), _p(inp['thres']), float(np.float32(self.image_scale)),
                                     _p(inp['feat_scale']), _p(grid), _p(mask), _p(pinv), None, _p(inp.get('aug_inv')),
                                     _ops._stream())
        _lib.check(rc, "df3d_project_voxels")
        return grid, mask, pinv

    def _canvas(self, x, grid, mask, pinv, inp):
        lib = _lib.load()
        feats = x.features.contiguous()
        n, C = feats.shape
        NI = inp['B'] * inp['ncam']
        winner = torch.empty((NI, inp['h'], inp['w']), dtype=torch.int32, device=feats.device)
        canvas = torch.empty((NI, C + 3, inp['h'], inp['w']), dtype=torch.float32, device=feats.device)
        rc = lib.df3d_scatter_to_image(_p(feats), _p(pinv), _p(x.indices.contiguous()), _p(grid), _p(mask), n, C,
                                       inp['B'], inp['ncam'], inp['h'], inp['w'], _p(winner), _p(canvas),
                                       _ops._stream())
        _lib.check(rc, "df3d_scatter_to_image")
        return canvas

    # ------------------------------------------------------------------ forward
    def _query_slots(self, ind, mask, B):
        """slot of every visible voxel inside its (sample, camera) list + max list length (one host sync)."""
        n = ind.shape[0]
        ncam = mask.shape[0]
        pos = torch.empty((ncam, n), dtype=torch.int32, device=ind.device)
        counts = torch.empty((B * ncam,), dtype=torch.int32, device=ind.device)
        rc = _lib.load().df3d_query_slots(_p(mask), _p(ind), n, B, ncam, _p(pos), _p(counts), _ops._stream())
        _lib.check(rc, "df3d_query_slots")
        max_ne = int(counts.max().item()) if n > 0 else 0                                     # the one host sync
        return pos, max_ne, counts

    def forward(self, batch_dict, example, encoded_voxel_list=None, layer_name=None, img_conv_func=None,
                fuse_mode=None, d_factor_list=None):
        if fuse_mode != 'pfat':
            raise NotImplementedError("fuse_mode %r" % (fuse_mode,))
        if torch.is_grad_enabled() and (encoded_voxel_list[-1].features.requires_grad
                                        or (self.training and any(p.requires_grad for p in self.parameters()))):
            # the inference formulation below (folded gate matrices, fused ACTR layers) records no autograd graph: a
            # training step takes the differentiable composition instead, so that fusion.pfat.* / fusion.ifat.* and the
            # voxel features keep learning
            return self.forward_autograd(batch_dict, example, encoded_voxel_list, layer_name, img_conv_func, d_factor_list)
        with torch.no_grad():
            return self._forward_inference(batch_dict, example, encoded_voxel_list, layer_name, img_conv_func, d_factor_list)

    def forward_autograd(self, batch_dict, example, encoded_voxel_list, layer_name=None, img_conv_func=None,
                         d_factor_list=None):
        """Training formulation of forward(): the integer work (projection with the reference's truncations, the winning
        voxel of every pixel, the per-camera query slots) comes from the same native kernels as in inference; every
        floating-point stage is a differentiable torch composition over them -- the image gate on gathered canvases
        (attention.py:31-61), ACTR through its module path (MSDeformAttnFunction backward = csrc/msda.hip), the additive
        write-back in camera order -- so gradients reach the voxel features of every scale the gate reads, the last
        scale's features, and all fusion parameters.  Same values as the inference path up to summation order."""
        x_last = encoded_voxel_list[-1]
        dev = x_last.features.device
        # ---- integer work: it depends on the voxel COORDINATES and the calibration only.  When the rulebooks of this step were
        #      built on a geometry stream (training on resident inputs, backbones._stem) it runs there too, so its host round
        #      trips (three index lists, the longest camera list) wait for a few small kernels instead of for everything the
        #      caller's stream still holds -- the previous step's backward included ----
        from .spconv.conv import GEOMETRY_STREAM_KEY
        geo = x_last.indice_dict.get(GEOMETRY_STREAM_KEY) if dev.type == "cuda" else None
        main = torch.cuda.current_stream(dev) if dev.type == "cuda" else None
        made = []

        def keep(*ts):
            made.extend(t for t in ts if torch.is_tensor(t))
            return ts[0] if len(ts) == 1 else ts

        with torch.no_grad(), (torch.cuda.stream(geo) if geo is not None else contextlib.nullcontext()):
            inp = self._gather_inputs(batch_dict, layer_name, dev)
            B, ncam, H, W = inp['B'], inp['ncam'], inp['h'], inp['w']
            NI = B * ncam
            last = len(encoded_voxel_list) - 1
            need = set([last]) | (set(self.ifat.voxel_idx) if self.ifat_cfg is not None else set())
            proj = {s_: keep(*self._project(encoded_voxel_list[s_], d_factor_list[s_], inp)) for s_ in sorted(need)}
            winners = {}
            if self.ifat_cfg is not None:
                for sidx in self.ifat.voxel_idx:
                    grid_s, mask_s, pinv_s = proj[sidx]
                    winner = self._winner(encoded_voxel_list[sidx], grid_s, mask_s, inp)    # [NI, H, W] row index or -1
                    img_w, pix_w = torch.nonzero(winner.view(NI, H * W) >= 0, as_tuple=True)
                    row_w = winner.view(NI, H * W)[img_w, pix_w].long()
                    winners[sidx] = keep(img_w, pix_w, row_w)
            grid, mask, pinv = proj[last]
            ind = x_last.indices.contiguous()
            pos, max_ne, counts = self._query_slots(ind, mask, B)
            # (camera, row) pairs that own a query slot, camera-major
            cam_i, row_i = torch.nonzero((mask != 0) & (pos.long() < max_ne), as_tuple=True)
            img_i = ind[row_i, 0].long() * ncam + cam_i
            slot_i = pos[cam_i, row_i].long()
            gx, gy = grid[cam_i, row_i, 0].long(), grid[cam_i, row_i, 1].long()
            qg = torch.stack([gx.to(torch.float32) / float(W), gy.to(torch.float32) / float(H)], 1)
            per_cam = torch.bincount(cam_i, minlength=ncam).tolist()        # camera-major lists: contiguous ranges
            keep(pos, counts, cam_i, row_i, img_i, slot_i, gx, gy, qg)
        if geo is not None:
            main.wait_stream(geo)
            for t in made:
                t.record_stream(main)
        imgs = _stack_views(inp['imgs'])                                            # [NI, Ci, H, W]
        if img_conv_func is not None:
            imgs = img_conv_func(imgs)
        gated = imgs
        att = None
        # The gate is one scalar per pixel and input_proj's convolution is 1x1, so input_proj(img * att) = att * (W img) + b and
        # the image feature at a query pixel is img[pixel] * att[pixel]: the gated copy of the camera maps (246 MB), its
        # gradient, the data-gradient GEMM back to it and the channel reduction of (gradient x image) that autograd ran for
        # d(att) never exist -- the maps themselves take no gradient (a frozen 2-D network produces them), d(att) comes from
        # the 128-channel projection.  DF3D_TRAIN_GATED=1: the composition over the gated maps (the reference's order).
        in_conv = self.pfat.input_proj[0][0]
        factored = (self.ifat_cfg is not None and os.environ.get("DF3D_TRAIN_GATED", "0") != "1" and not imgs.requires_grad
                    and tuple(in_conv.kernel_size) == (1, 1) and tuple(in_conv.stride) == (1, 1) and in_conv.groups == 1
                    and len(self.pfat.input_proj) == 1 and self.pfat.feature_modal in ('image', 'hybrid'))
        if self.ifat_cfg is not None and factored and os.environ.get("DF3D_TRAIN_GATE_SPARSE", "1") == "1":
            # the gate from the voxel rows themselves (no [NI, H W, C + 3] canvases: Basicgate...attention_sparse)
            rows = {sidx: torch.cat([encoded_voxel_list[sidx].features, proj[sidx][2]], 1) for sidx in self.ifat.voxel_idx}
            att = self.ifat.attention_sparse(imgs, rows, winners)
        elif self.ifat_cfg is not None:
            canvases = {}
            for sidx in self.ifat.voxel_idx:
                x = encoded_voxel_list[sidx]
                img_w, pix_w, row_w = winners[sidx]
                rows = torch.cat([x.features, proj[sidx][2]], 1)                    # [n, C + 3]
                # pts2img: the winner's row at its pixel, zero elsewhere (a gather over the occupied pixels only: the
                # backward of a dense gather with every empty pixel clamped to one row serialises on that row)
                canvases[sidx] = rows.new_zeros((NI, H * W, rows.shape[1])).index_put((img_w, pix_w), rows[row_w])
            if factored:
                att = self.ifat.attention_rows(imgs, canvases)                      # [NI, H, W]
            else:
                gated = self.ifat.forward_rows(imgs, canvases)
        feats = x_last.features
        n, C = feats.shape
        Ci = gated.shape[1]
        v_feat = feats.new_zeros((NI, max_ne, C)).index_put((img_i, slot_i), feats[row_i])
        qgrid = feats.new_zeros((NI, max_ne, 2)).index_put((img_i, slot_i), qg.to(feats.dtype))
        qpts = feats.new_zeros((NI, max_ne, 3)).index_put((img_i, slot_i), pinv[row_i])
        if att is not None:
            rows_i = imgs[img_i, :, gy, gx] * att[img_i, gy, gx].unsqueeze(1)
            v_i_feat = feats.new_zeros((NI, max_ne, Ci)).index_put((img_i, slot_i), rows_i)
            u = _ops.channel_first_linear(imgs.reshape(NI, Ci, H * W), in_conv.weight[:, :, 0, 0])   # [NI, C', H*W]
            src_conv = u * att.reshape(NI, 1, H * W)
            if in_conv.bias is not None:
                src_conv = src_conv + in_conv.bias[None, :, None]
            enh = self.pfat.forward_projected(v_feat, qgrid, src_conv.view(NI, -1, H, W), v_i_feat, qpts)
        else:
            v_i_feat = feats.new_zeros((NI, max_ne, Ci)).index_put((img_i, slot_i), gated[img_i, :, gy, gx])
            enh = self.pfat(v_feat, qgrid, [gated], v_i_feat, qpts)                     # [NI, max_ne, C]
        out = feats
        a = 0
        for cam in range(ncam):                                                     # additive, camera order (writeback_kernel)
            b = a + int(per_cam[cam])
            if b > a:
                out = out.index_add(0, row_i[a:b], enh[img_i[a:b], slot_i[a:b]])
            a = b
        return x_last.replace_feature(out)

    def _winner(self, x, grid, mask, inp):
        """Row index of the voxel that owns each image pixel (-1: none): the pts2img scatter's last writer."""
        n = x.features.shape[0]
        NI = inp['B'] * inp['ncam']
        winner = torch.empty((NI, inp['h'], inp['w']), dtype=torch.int32, device=x.features.device)
        rc = _lib.load().df3d_scatter_winner(_p(x.indices.contiguous()), _p(grid), _p(mask), n, inp['B'], inp['ncam'],
                                             inp['h'], inp['w'], _p(winner), _ops._stream())
        _lib.check(rc, "df3d_scatter_winner")
        return winner

    def _forward_inference(self, batch_dict, example, encoded_voxel_list, layer_name, img_conv_func, d_factor_list):
        lib = _lib.load()
        x_last = encoded_voxel_list[-1]
        dev = x_last.features.device
        pre = self._prefetched.pop((id(batch_dict), layer_name), None)
        if pre is not None and pre[3] is not batch_dict:
            pre = None
        prep, self._prepared = self._prepared, None
        prep = prep[2] if (prep is not None and prep[0] == id(batch_dict) and prep[1] == layer_name) else None
        if pre is not None:
            inp, both, ev = pre[:3]
            if ev is not None:                       # produced on the side stream
                main = torch.cuda.current_stream(dev)
                main.wait_event(ev)
                for t_ in (both if isinstance(both, tuple) else (both,)):
                    t_.record_stream(main)
        elif prep is not None and "both" in prep and img_conv_func is None:
            inp, both = prep['inp'], prep['both']          # projected by the frame-head worker (the caller's stream waited)
        else:
            inp = prep['inp'] if prep is not None else self._gather_inputs(batch_dict, layer_name, dev)
            both = self._image_projection(inp, img_conv_func,
                                          pixrow=prep.get("pixrow") if (prep is not None and img_conv_func is None) else None)
        B, ncam = inp['B'], inp['ncam']
        NI = B * ncam
        H, W = inp['h'], inp['w']
        last = len(encoded_voxel_list) - 1
        # (a7) projection of every scale the gate or the queries need
        need = set([last])
        if self.ifat_cfg is not None:
            need |= set(self.ifat.voxel_idx)
        proj = {}
        early = None
        if prep is not None:
            proj, early = dict(prep['proj']), prep['early']
        elif getattr(x_last, "_indices_synced", False) and os.environ.get("DF3D_EARLY_SLOTS", "1") == "1":
            # The query lists depend on the voxel COORDINATES only, and those are complete (the host has read their
            # count).  Project + count on a side stream now: the one host round trip of this adapter (max list
            # length) then does not wait for the convolutions still queued on the main stream, and the host can
            # enqueue the rest of the adapter while they run.
            if self._side is None:
                self._side = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream(dev)
            with torch.cuda.stream(self._side):
                proj[last] = self._project(x_last, d_factor_list[last], inp)
                early = self._query_slots(x_last.indices.contiguous(), proj[last][1], B)
            main.wait_stream(self._side)
            for t in proj[last] + (early[0], early[2]):
                t.record_stream(main)
        for s_ in sorted(need):
            if s_ not in proj:
                proj[s_] = self._project(encoded_voxel_list[s_], d_factor_list[s_], inp)
        in_conv = self.pfat.input_proj[0][0]
        att = None
        S_pix = H * W
        w_ip = in_conv.weight[:, :, 0, 0]
        if self.ifat_cfg is not None:
            # (a9) image-side gate, canvas-free: the projection above also produced the gate's 1-channel image
            # summary (extra GEMM row); the voxel side is 9 scalars per visible voxel
            T, kg, w3, b3 = self.ifat.folded()
            fused_tail = os.environ.get("DF3D_FUSION_TAIL", "1") == "1"
            gate = both[1] if isinstance(both, tuple) else both[:, w_ip.shape[0]]                # [NI, H*W]
            gate_bias = b3 if (fused_tail and gate.is_contiguous() and b3.numel() == 1 and b3.dtype == torch.float32) else None
            if gate_bias is None:
                gate = gate + b3
            S = torch.empty((NI, 9, H, W), dtype=torch.float32, device=dev)
            winner = None
            have = prep.get('winner', {}) if (prep is not None and fused_tail) else {}
            first = True
            for sidx in self.ifat.voxel_idx:
                x = encoded_voxel_list[sidx]
                grid_s, mask_s, pinv_s = proj[sidx]
                feats_s, Ts = x.features, T[sidx]
                if (sidx in have and feats_s.dtype == torch.float32 and feats_s.shape[1] % 4 == 0 and Ts.dtype == torch.float32
                        and os.environ.get("DF3D_GATE_ROWS", "1") == "1"):
                    # the winner map of this scale came with the frame head (it depends on the coordinates alone)
                    rc = lib.df3d_gate_rows(_p(feats_s.contiguous()), feats_s.shape[1], _p(pinv_s.contiguous()),
                                            _p(Ts.contiguous()), _p(have[sidx]), NI, H, W, _p(S), int(first), _ops._stream())
                    _lib.check(rc, "df3d_gate_rows")
                    first = False
                    continue
                if winner is None:
                    winner = torch.empty((NI, H, W), dtype=torch.int32, device=dev)
                if (feats_s.dtype == torch.float32 and feats_s.shape[1] % 4 == 0 and Ts.dtype == torch.float32
                        and os.environ.get("DF3D_GATE_ROWS", "1") == "1"):
                    # the 9 responses of a row matter for the rows that win a pixel only: computed inside the scatter
                    rc = lib.df3d_gate_scatter_rows(_p(feats_s.contiguous()), feats_s.shape[1], _p(pinv_s.contiguous()),
                                                    _p(Ts.contiguous()), _p(x.indices.contiguous()), _p(grid_s), _p(mask_s),
                                                    feats_s.shape[0], B, ncam, H, W, _p(winner), _p(S), int(first),
                                                    _ops._stream())
                    _lib.check(rc, "df3d_gate_scatter_rows")
                else:
                    rows = torch.cat([feats_s, pinv_s], 1)
                    s9 = (rows @ Ts.t()).contiguous()                           # [n, 9]
                    rc = lib.df3d_gate_scatter(_p(s9), _p(x.indices.contiguous()), _p(grid_s), _p(mask_s), rows.shape[0],
                                               B, ncam, H, W, _p(winner), _p(S), int(first), _ops._stream())
                    _lib.check(rc, "df3d_gate_scatter")
                first = False
            att = torch.empty((NI, H, W), dtype=torch.float32, device=dev)
            if gate_bias is not None:                       # the summary's bias is added inside (no element-wise pass)
                rc = lib.df3d_gate_finish_bias(_p(gate), _p(gate_bias), _p(S), _p(kg), NI, H, W, _p(att), _ops._stream())
            else:
                rc = lib.df3d_gate_finish(_p(gate), _p(S), _p(kg), NI, H, W, _p(att), _ops._stream())
            _lib.check(rc, "df3d_gate_finish")
        fold = self.pfat.can_fold()
        if not fold:
            # input_proj(img * att) = att * (W img) + b   (att is a per-pixel scalar)
            src_conv = both[:, :w_ip.shape[0]]
            if att is not None:
                src_conv = src_conv * att.view(NI, 1, S_pix)
            src_conv = (src_conv + in_conv.bias[None, :, None]).view(NI, -1, H, W)
        values = None
        u_rows = both[0] if isinstance(both, tuple) else both
        if (fold and u_rows.dtype == torch.uint8 and os.environ.get("DF3D_VALUE_SIDE", "0") == "1"):
            # the image side of ACTR (370 MB streamed) beside the query assembly and the query-side small launches
            values = self.pfat.start_values(u_rows, None if att is None else att.view(NI, S_pix))
        # (a8) per-camera query sets from the LAST scale (Appendix C item 4)
        grid, mask, pinv = proj[last]
        feats = x_last.features.contiguous()
        ind = x_last.indices.contiguous()
        n, C = feats.shape
        pos, max_ne, counts = early if early is not None else self._query_slots(ind, mask, B)
        Ci = inp['Ci']
        v_feat = torch.empty((NI, max_ne, C), dtype=torch.float32, device=dev)
        v_i_feat = torch.empty((NI, max_ne, Ci), dtype=torch.float32, device=dev)
        qgrid = torch.empty((NI, max_ne, 2), dtype=torch.float32, device=dev)
        qpts = torch.empty((NI, max_ne, 3), dtype=torch.float32, device=dev)
        depth_pos = self.pfat.pos_encode_method == "depth"
        qpos = torch.empty((NI, max_ne, C), dtype=torch.float32, device=dev) if depth_pos else None
        compact = both[2] if (isinstance(both, tuple) and len(both) > 2 and prep is not None and "pixrow" in prep) else None
        by_slot = (counts is not None and NI < 12 and C <= 256 and Ci <= 512
                   and os.environ.get("DF3D_ASSEMBLE_SLOTS", "1") == "1")
        if by_slot:
            # a wave per SLOT (no dead candidate waves); with `compact` the image rows of the sampled pixels are pixel-major
            # (written by the image projection): one contiguous 1 KB row per query instead of 256 scattered elements
            slot_rows = torch.empty((NI * max_ne, 4), dtype=torch.int32, device=dev)
            rc = lib.df3d_assemble_queries2_slots(_p(feats), _p(pinv), _p(ind), _p(grid), _p(mask), _p(pos), None,
                                                  _p(inp['img_ptrs']), _p(att), n, C, Ci, B, ncam, H, W, max_ne, _p(v_feat),
                                                  _p(v_i_feat), _p(qgrid), _p(qpts), _p(qpos), _p(counts), _p(slot_rows),
                                                  _p(prep["pixrow"][0]) if compact is not None else None, _p(compact),
                                                  _ops._stream())
            _lib.check(rc, "df3d_assemble_queries2_slots")
        elif compact is not None:
            rc = lib.df3d_assemble_queries2_compact(_p(feats), _p(pinv), _p(ind), _p(grid), _p(mask), _p(pos), _p(prep["pixrow"][0]),
                                                    _p(compact), _p(att), n, C, Ci, B, ncam, H, W, max_ne, _p(v_feat),
                                                    _p(v_i_feat), _p(qgrid), _p(qpts), _p(qpos), _p(counts), _ops._stream())
            _lib.check(rc, "df3d_assemble_queries2_compact")
        else:
            rc = lib.df3d_assemble_queries2(_p(feats), _p(pinv), _p(ind), _p(grid), _p(mask), _p(pos), None, _p(inp['img_ptrs']),
                                            _p(att), n, C, Ci, B, ncam, H, W, max_ne, _p(v_feat), _p(v_i_feat), _p(qgrid),
                                            _p(qpts), _p(qpos), _p(counts), _ops._stream())
            _lib.check(rc, "df3d_assemble_queries2")
        # (a10-a12) ACTR
        if fold:
            enh = self.pfat.forward_folded(v_feat, qgrid, both[0] if isinstance(both, tuple) else both,
                                           None if att is None else att.view(NI, S_pix), (H, W),
                                           v_i_feat, qpts, q_pos=qpos, values=values).contiguous()
        else:
            enh = self.pfat.forward_projected(v_feat, qgrid, src_conv, v_i_feat, qpts, q_pos=qpos).contiguous()
        # write-back, additive, camera order (Appendix C item 8)
        out = torch.empty_like(feats)
        if C % 8 == 0 and os.environ.get("DF3D_FUSION_TAIL", "1") == "1":
            # 16-byte accesses + the operand split the convolution behind the adapter reads (no df3d_split_rows pass)
            osplit = torch.empty((n, 4 * C), dtype=torch.uint8, device=dev) if _ops.CONV_PRECISION == "split" else None
            rc = lib.df3d_fusion_writeback_split(_p(feats), _p(enh), _p(ind), _p(mask), _p(pos), n, C, ncam, max_ne, _p(out),
                                                 _p(osplit), _ops._stream())
            _lib.check(rc, "df3d_fusion_writeback_split")
            y = x_last.replace_feature(out)
            if osplit is not None:
                y._split = (out, osplit)
            return y
        rc = lib.df3d_fusion_writeback(_p(feats), _p(enh), _p(ind), _p(mask), _p(pos), n, C, ncam, max_ne, _p(out),
                                       _ops._stream())
        _lib.check(rc, "df3d_fusion_writeback")
        return x_last.replace_feature(out)


# ---------------------------------------------------------------------- config-2 helpers (bench / tests)
CP_DEPTH_THRES = {'CAM_FRONT': 1, 'CAM_FRONT_LEFT': 0, 'CAM_FRONT_RIGHT': 0, 'CAM_BACK': 0.5, 'CAM_BACK_LEFT': 0,
                  'CAM_BACK_RIGHT': 0}
CP_PFAT_CFG = dict(fusion_method='sum', feature_modal='hybrid',
                   hybrid_cfg=dict(attn_layer='BiGateSum1D_2', q_method='sum', q_rep_place=['weight']), num_bins=80,
                   num_channels=[256], query_num_feat=128, num_enc_layers=2, max_num_ne_voxel=26000,
                   pos_encode_method='depth')
CP_LT_CFG = dict(npoint=2048, radius=2.0, nsample=32, num_layers=2, attn_feat_agg_method='unique',
                 feat_agg_method='replace')
CP_IFAT_CFG = dict(fusion_method='Basicgate_patch_iv_multivoxel', img_num_channel=256, pts_num_channel=128,
                   voxel_feat_channel=[32, 64, 128], voxel_idx=[0, 2])


def build_centerpoint_fusion(voxel_size=synth.NUSC_VOXEL, pc_range=synth.NUSC_RANGE, image_scale=2.0 / 3.0):
    """The `fusion=dict(type='VoxelWithPointProjection', ...)` block of
    CP/configs/nusc/voxelnet/nusc_centerpoint_voxelnet_0075voxel_fix_bn_z_multimodal_pfat_hybrid7_ifat.py:71-109."""
    return VoxelWithPointProjection(fuse_mode='pfat', interpolate=False, voxel_size=voxel_size, pc_range=pc_range,
                                    image_list=synth.NUSC_CAMS, image_scale=image_scale, depth_thres=CP_DEPTH_THRES,
                                    pfat_cfg=dict(CP_PFAT_CFG), lt_cfg=dict(CP_LT_CFG), ifat_cfg=dict(CP_IFAT_CFG),
                                    model_name='ACTR')


def synthetic_camera_inputs(batch, dev, seed=1234, raw_hw=(900, 1600), image_scale=2.0 / 3.0, feat_hw=(150, 267),
                            yaw_offset_deg=0.0):
    """batch_dict / example with the keys the adapter reads: six synthetic
    DeepLabV3-layer1-shaped feature maps [B,256,150,267] (N(0,1)), calibration of six pinhole cameras
    at 60 deg spacing, scaled image shape (600, 1067)."""
    cams = synth.nusc_cameras(image_hw=raw_hw, yaw_offset_deg=yaw_offset_deg)
    H, W = int(round(raw_hw[0] * image_scale)), int(round(raw_hw[1] * image_scale))
    feats = synth.camera_features(batch * 6, 256, feat_hw, seed).reshape(batch, 6, 256, feat_hw[0], feat_hw[1])
    # the camera network's output for the B*6 images is one tensor; the reference hands it on as a dict of
    # per-camera slices (CP/det3d/models/detectors/voxelnet.py) -- same here: views, not copies
    feats_dev = torch.from_numpy(np.ascontiguousarray(feats)).to(dev)
    batch_dict = {'image_shape': {}, 'img_feat': {'layer1_ori_feat2d': {}}, 'calib': {}}
    for i, name in enumerate(synth.NUSC_CAMS):
        key = name.lower()
        batch_dict['image_shape'][key] = torch.tensor([[H, W, 3]] * batch)
        batch_dict['img_feat']['layer1_ori_feat2d'][key] = feats_dev[:, i]
        T, K = cams[name]
        ck = key.lstrip('cam_')
        batch_dict['calib']['lidar2cam_' + ck] = torch.from_numpy(np.stack([T] * batch)).to(dev)
        batch_dict['calib']['cam_intrinsic_' + ck] = torch.from_numpy(np.stack([K] * batch)).to(dev)
    return batch_dict, {}
