"""Oracle compositions (numpy, CPU) of the reference's backbone forward passes, driven by a
module state_dict.  Test infrastructure only: used by the GPU parity tests, smoke() and
bench.py's cpu_baseline leg.  Follows CP/det3d/models/backbones/scn.py:51-94,112-160,200-236."""
import numpy as np

from oracle import oracle as orc

BN_EPS = 1e-3


def _bn(sd, prefix, x):
    return orc.batchnorm_eval(x, sd[prefix + ".weight"], sd[prefix + ".bias"], sd[prefix + ".running_mean"],
                              sd[prefix + ".running_var"], BN_EPS)


class SpTensor(object):
    def __init__(self, features, indices, shape, batch, rulebooks=None):
        self.features, self.indices, self.shape, self.batch = features, indices, list(shape), batch
        self.rulebooks = rulebooks if rulebooks is not None else {}


# the two index / convolution providers a composition can run on: the restatement (`oracle.oracle`) or the
# reference's own compiled CPU code (`oracle.ref`, oracle/_ref/*.so); both expose get_indice_pairs / indice_conv
_IMPL = [orc]


class using(object):
    """with using(ref): ... -- run the compositions below on another provider (e.g. the reference's compiled CPU ops)."""

    def __init__(self, impl):
        self.impl = impl

    def __enter__(self):
        _IMPL.append(self.impl)

    def __exit__(self, *exc):
        _IMPL.pop()


STAGE_SECONDS = {}      # filled when timing is on: {'rulebook': s, 'conv': s}


def _conv(sd, prefix, x, ks, stride, padding, subm, key=None):
    import time
    impl = _IMPL[-1]
    w = sd[prefix + ".weight"]
    rb = x.rulebooks.get(key) if key is not None else None
    t0 = time.perf_counter()
    if rb is None:
        rb = impl.get_indice_pairs(x.indices, x.batch, x.shape, ks, stride, padding, [1, 1, 1], subm)
        if key is not None:
            x.rulebooks[key] = rb
    outids, pairs, num, oshape = rb
    t1 = time.perf_counter()
    y = impl.indice_conv(x.features, w, pairs, num, len(outids), subm)
    t2 = time.perf_counter()
    STAGE_SECONDS["rulebook"] = STAGE_SECONDS.get("rulebook", 0.0) + (t1 - t0)
    STAGE_SECONDS["conv"] = STAGE_SECONDS.get("conv", 0.0) + (t2 - t1)
    if prefix + ".bias" in sd:
        y = y + sd[prefix + ".bias"]
    return SpTensor(y.astype(np.float32), outids, oshape, x.batch, x.rulebooks)


def _basic_block(sd, prefix, x, key):
    out = _conv(sd, prefix + ".conv1", x, [3, 3, 3], [1, 1, 1], [1, 1, 1], 1, key)
    out.features = np.maximum(_bn(sd, prefix + ".bn1", out.features), 0)
    out = _conv(sd, prefix + ".conv2", out, [3, 3, 3], [1, 1, 1], [1, 1, 1], 1, key)
    out.features = _bn(sd, prefix + ".bn2", out.features)
    out.features = np.maximum(out.features + x.features, 0)
    return out


def _down(sd, prefix, x, ks, stride, padding):
    out = _conv(sd, prefix + ".0", x, ks, stride, padding, 0)
    out.features = np.maximum(_bn(sd, prefix + ".1", out.features), 0)
    return out


def centerpoint_backbone(sd, voxel_features, coors, batch_size, input_shape, fuse=None):
    """SpMiddleResNetFHD(.Fusion).forward.  sd: numpy state_dict.  Returns (dense [B,256,H,W], convs dict)."""
    shape = list(np.array(input_shape[::-1]) + [1, 0, 0])
    x = SpTensor(np.asarray(voxel_features, np.float32), np.asarray(coors, np.int32), shape, batch_size)
    x = _conv(sd, "conv_input.0", x, [3, 3, 3], [1, 1, 1], [1, 1, 1], 1, "res0")
    x.features = np.maximum(_bn(sd, "conv_input.1", x.features), 0)
    c1 = _basic_block(sd, "conv1.1", _basic_block(sd, "conv1.0", x, "res0"), "res0")
    t = _down(sd, "conv2", c1, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    c2 = _basic_block(sd, "conv2.4", _basic_block(sd, "conv2.3", t, "res1"), "res1")
    t = _down(sd, "conv3", c2, [3, 3, 3], [2, 2, 2], [1, 1, 1])
    c3 = _basic_block(sd, "conv3.4", _basic_block(sd, "conv3.3", t, "res2"), "res2")
    t = _down(sd, "conv4", c3, [3, 3, 3], [2, 2, 2], [0, 1, 1])
    c4 = _basic_block(sd, "conv4.4", _basic_block(sd, "conv4.3", t, "res3"), "res3")
    if fuse is not None:
        c4 = fuse(c2, c3, c4)
    e = _down(sd, "extra_conv", c4, [3, 1, 1], [2, 1, 1], [0, 0, 0])
    d = orc.dense(e.features, e.indices, e.shape, batch_size)
    B, C, D, H, W = d.shape
    return d.reshape(B, C * D, H, W), {"conv1": c1, "conv2": c2, "conv3": c3, "conv4": c4}


def sort_rows(indices, features):
    o = np.lexsort((indices[:, 3], indices[:, 2], indices[:, 1], indices[:, 0]))
    return indices[o], features[o]


# ------------------------------------------------------------------------- LocalTransformer (a13)
def local_transformer(sd, xyz, feat, npoint, radius, nsample, nhead=4, num_layers=2, winner="first"):
    """CP/det3d/models/model_utils/pointformer.py:349-380 restated on the CPU (oracle index ops +
    torch-CPU fp32 for the dense layers).  sd: numpy state_dict of LocalTransformer.  xyz [B,N,3],
    feat [B,C,N] -> [B,N,C].  `winner`: which duplicate the 'unique' scatter keeps (flat position)."""
    import torch
    import torch.nn.functional as F
    t = lambda k: torch.from_numpy(np.ascontiguousarray(sd[k]))
    B, C, N = feat.shape
    fps = orc.furthest_point_sample(xyz, npoint)
    new_xyz = np.stack([xyz[b][fps[b]] for b in range(B)])
    idx = orc.ball_query(0.0, radius, nsample, xyz, new_xyz)
    gx = torch.from_numpy(orc.group_points(np.ascontiguousarray(xyz.transpose(0, 2, 1)), idx))
    gf = torch.from_numpy(orc.group_points(feat, idx))
    pe = F.conv2d(gx, t("pe.0.conv.weight"))
    pe = F.batch_norm(pe, t("pe.0.bn.running_mean"), t("pe.0.bn.running_var"), t("pe.0.bn.weight"), t("pe.0.bn.bias"),
                      False, 0.0, 1e-5)
    pe = F.conv2d(F.relu(pe), t("pe.1.conv.weight"), t("pe.1.conv.bias"))
    x = gf + pe
    D, ns = C, nsample
    x = x.permute(0, 2, 1, 3).reshape(-1, D, ns).permute(2, 0, 1)        # [ns, B*np, D]
    for j in range(num_layers):
        p = "chunk.layers.%d." % j
        h = F.layer_norm(x, (D,), t(p + "norm1.weight"), t(p + "norm1.bias"))
        a, _ = F.multi_head_attention_forward(h, h, h, D, nhead, t(p + "self_attn.in_proj_weight"),
                                              t(p + "self_attn.in_proj_bias"), None, None, False, 0.0,
                                              t(p + "self_attn.out_proj.weight"), t(p + "self_attn.out_proj.bias"),
                                              training=False, need_weights=False)
        h = h + a
        h2 = F.layer_norm(h, (D,), t(p + "norm2.weight"), t(p + "norm2.bias"))
        f = F.linear(F.relu(F.linear(h2, t(p + "linear1.weight"), t(p + "linear1.bias"))), t(p + "linear2.weight"),
                     t(p + "linear2.bias"))
        x = h2 + f
    y = x.permute(1, 2, 0).reshape(B, npoint, D, ns).transpose(1, 2).numpy()   # [B,D,np,ns]
    out = feat.copy()
    for b in range(B):
        idf = idx[b].reshape(-1)
        ff = y[b].reshape(C, -1)
        order = range(len(idf) - 1, -1, -1) if winner == "first" else range(len(idf))
        for pos in order:               # later writes win
            out[b][:, idf[pos]] = ff[:, pos]
    return out.transpose(0, 2, 1)


# ------------------------------------------------------------------------- TransFusion / Voxel-RCNN encoders
def _cbr(sd, pconv, pbn, x, ks, stride, padding, subm, key=None):
    out = _conv(sd, pconv, x, ks, stride, padding, subm, key)
    out.features = np.maximum(_bn(sd, pbn, out.features), 0)
    return out


def transfusion_encoder(sd, voxel_features, coors, batch_size, sparse_shape, encoder_channels, encoder_paddings,
                        fuse=None, fusion_pos=None):
    """TF/mmdet3d/models/middle_encoders/sparse_encoder.py:321-372 with block_type='basicblock'
    (BasicBlock = TF/mmdet3d/ops/sparse_block.py:67-120: convs without bias, no indice_key)."""
    x = SpTensor(np.asarray(voxel_features, np.float32), np.asarray(coors, np.int32), sparse_shape, batch_size)
    x = _cbr(sd, "conv_input.0", "conv_input.1", x, [3, 3, 3], [1, 1, 1], [1, 1, 1], 1, "subm1")
    for i, blocks in enumerate(encoder_channels):
        for j, _ in enumerate(blocks):
            p = "encoder_layers.encoder_layer%d.%d" % (i + 1, j)
            if j == len(blocks) - 1 and i != len(encoder_channels) - 1:
                pad = encoder_paddings[i][j]
                pad = list(pad) if isinstance(pad, (list, tuple)) else [pad] * 3
                x = _cbr(sd, p + ".0", p + ".1", x, [3, 3, 3], [2, 2, 2], pad, 0)
            else:
                idn = x.features
                o = _conv(sd, p + ".conv1", x, [3, 3, 3], [1, 1, 1], [1, 1, 1], 1, ("s", i))
                o.features = np.maximum(_bn(sd, p + ".bn1", o.features), 0)
                o = _conv(sd, p + ".conv2", o, [3, 3, 3], [1, 1, 1], [1, 1, 1], 1, ("s", i))
                o.features = np.maximum(_bn(sd, p + ".bn2", o.features) + idn, 0)
                x = o
        if fuse is not None and fusion_pos is not None and i in fusion_pos:
            x = fuse(x)
    e = _cbr(sd, "conv_out.0", "conv_out.1", x, [3, 1, 1], [2, 1, 1], [0, 0, 0], 0)
    d = orc.dense(e.features, e.indices, e.shape, batch_size)
    B, C, D, H, W = d.shape
    return d.reshape(B, C * D, H, W), x


def voxel_backbone8x(sd, voxel_features, coors, batch_size, sparse_shape, fuse1=None, fuse4=None):
    """VR/pcdet/models/backbones_3d/spconv_backbone.py:135-243 (LiDAR branch); with `fuse1` / `fuse4` the fusion variant's
    order of operations (:829-916): fuse1(x_conv1) right behind conv1, fuse4(x_conv2, x_conv3, x_conv4) right behind conv4,
    both in front of whatever reads those tensors next."""
    x = SpTensor(np.asarray(voxel_features, np.float32), np.asarray(coors, np.int32), sparse_shape, batch_size)
    x = _cbr(sd, "conv_input.0", "conv_input.1", x, [3, 3, 3], [1, 1, 1], [1, 1, 1], 1, "subm1")
    c1 = _cbr(sd, "conv1.0.0", "conv1.0.1", x, [3, 3, 3], [1, 1, 1], [1, 1, 1], 1, "subm1")
    if fuse1 is not None:
        c1 = fuse1(c1)
    outs = {"x_conv1": c1}
    cur = c1
    for s, pad in ((2, [1, 1, 1]), (3, [1, 1, 1]), (4, [0, 1, 1])):
        cur = _cbr(sd, "conv%d.0.0" % s, "conv%d.0.1" % s, cur, [3, 3, 3], [2, 2, 2], pad, 0)
        for j in (1, 2):
            cur = _cbr(sd, "conv%d.%d.0" % (s, j), "conv%d.%d.1" % (s, j), cur, [3, 3, 3], [1, 1, 1], [1, 1, 1], 1,
                       "subm%d" % s)
        outs["x_conv%d" % s] = cur
    if fuse4 is not None:
        cur = outs["x_conv4"] = fuse4(outs["x_conv2"], outs["x_conv3"], cur)
    out = _cbr(sd, "conv_out.0", "conv_out.1", cur, [3, 1, 1], [2, 1, 1], [0, 0, 0], 0)
    return out, outs


# ------------------------------------------------------------------------- ACTR + CenterPoint fusion adapter (a7-a12)
def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a))


def actr_forward(sd, v_feat, grid, i_feat, lidar_grid, v_i_feat, num_layers=2, n_heads=8, n_points=4, prefix=""):
    """ACTR.forward with feature_modal='hybrid', pos_encode_method='depth', q_method 'sum' / rep_place
    ['weight'], BiGateSum1D_2 (CP/det3d/models/model_utils/actr.py:131-187, actr_transformer.py:399-426,
    473-511, ops/modules/ms_deform_attn.py:98-190, attentions.py:96-117, position_encoding.py:107-120).
    Dense layers in torch-CPU fp32, deformable sampling by the numpy oracle."""
    import torch
    import torch.nn.functional as F
    P = lambda k: _t(sd[prefix + k])
    v_feat, grid, i_feat, lidar_grid, v_i_feat = [_t(np.asarray(a, np.float32)) for a in
                                                  (v_feat, grid, i_feat, lidar_grid, v_i_feat)]
    N, Q, C = v_feat.shape
    H, W = i_feat.shape[2:]
    qi = F.conv1d(v_i_feat.transpose(1, 2), P("i_input_proj.0.weight"), P("i_input_proj.0.bias"))
    qi = F.group_norm(qi, 32, P("i_input_proj.1.weight"), P("i_input_proj.1.bias"), 1e-5).transpose(1, 2)
    d = lidar_grid[..., 0] / 60.0 * (2 * np.pi)
    dim_t = torch.arange(C, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / C)
    pd = d[:, :, None] / dim_t
    q_pos = torch.stack((pd[:, :, 0::2].sin(), pd[:, :, 1::2].cos()), dim=3).flatten(2)
    src = F.group_norm(F.conv2d(i_feat, P("input_proj.0.0.weight"), P("input_proj.0.0.bias")), 32,
                       P("input_proj.0.1.weight"), P("input_proj.0.1.bias"), 1e-5)
    src = src.flatten(2).transpose(1, 2)                                   # [N, HW, C]
    q = v_feat
    D = C // n_heads
    for i in range(num_layers):
        L = "transformer.encoder.layers.%d." % i
        value = F.linear(src, P(L + "self_attn.value_proj.weight"), P(L + "self_attn.value_proj.bias"))
        query, iq = q + q_pos, qi + q_pos
        off = F.linear(query, P(L + "self_attn.sampling_offsets.weight"), P(L + "self_attn.sampling_offsets.bias"))
        aw = F.linear(query + iq, P(L + "self_attn.attention_weights.weight"), P(L + "self_attn.attention_weights.bias"))
        aw = torch.softmax(aw.view(N, Q, n_heads, n_points), -1).view(N, Q, n_heads, 1, n_points)
        off = off.view(N, Q, n_heads, 1, n_points, 2)
        loc = grid[:, :, None, None, None, :] + off / torch.tensor([W, H], dtype=torch.float32)
        att = orc.ms_deform_attn(value.view(N, H * W, n_heads, D).numpy(), [(H, W)], loc.numpy(), aw.numpy())
        att = F.linear(_t(att), P(L + "self_attn.output_proj.weight"), P(L + "self_attn.output_proj.bias"))
        qi = F.layer_norm(qi + att, (C,), P(L + "norm1.weight"), P(L + "norm1.bias"))
        qi = F.layer_norm(qi + F.linear(F.relu(F.linear(qi, P(L + "linear1.weight"), P(L + "linear1.bias"))),
                                        P(L + "linear2.weight"), P(L + "linear2.bias")), (C,),
                          P(L + "norm2.weight"), P(L + "norm2.bias"))
        q = F.layer_norm(q + F.linear(F.relu(F.linear(q, P(L + "linear3.weight"), P(L + "linear3.bias"))),
                                      P(L + "linear4.weight"), P(L + "linear4.bias")), (C,),
                         P(L + "norm3.weight"), P(L + "norm3.bias"))
        fuse = (q + qi).transpose(1, 2)
        s1 = torch.sigmoid(F.conv1d(fuse, P(L + "fusion_layer.b_conv1d.weight"), P(L + "fusion_layer.b_conv1d.bias"))).transpose(1, 2)
        s2 = torch.sigmoid(F.conv1d(fuse, P(L + "fusion_layer.a_conv1d.weight"), P(L + "fusion_layer.a_conv1d.bias"))).transpose(1, 2)
        q, qi = q + qi * s1, qi + q * s2
    return q.numpy()


def centerpoint_fusion(sd, levels, img_feats, calib, image_hw, cams, voxel_size, pc_range, image_scale, depth_thres,
                       d_factors=(2, 4, 8), ifat_idx=(0, 2), debug=None):
    """VoxelWithPointProjection.forward, fuse_mode 'pfat' + ifat gate (CP/det3d/models/fusion/
    voxel_with_point_projection.py:131-385, point_to_image_projection.py:63-231,
    model_utils/attention.py:31-61,422-468) restated with explicit loops.
    levels: [(indices [n,4], features [n,C])] for x_conv2..4 (rows batch-sorted); img_feats {cam: [B,256,h,w]};
    calib {cam: (lidar2cam [B,4,4], intrinsic [B,3,3])}; image_hw (H, W) of the network input image.
    Returns the fused features of the last level."""
    import torch
    import torch.nn.functional as F
    P = lambda k: _t(sd[k])
    B = next(iter(img_feats.values())).shape[0]
    H, W = image_hw
    pc_min = torch.tensor(pc_range[:3], dtype=torch.float32)
    vs = torch.tensor(voxel_size, dtype=torch.float32)
    n_last = len(levels) - 1
    per = {}                    # (b, cam) -> dict
    img_gated = {}
    for ci, cam in enumerate(cams):
        l2c, K = [_t(np.asarray(x, np.float32)) for x in calib[cam]]
        proj = []
        for (ind, _), dfac in zip(levels, d_factors):
            ind_t = _t(ind).float()
            xyz = ind_t[:, [3, 2, 1]] * (vs * dfac) + pc_min                       # voxel corner
            out = []
            for b in range(B):
                sel = ind[:, 0] == b
                p = xyz[sel]
                ph = torch.cat([p, torch.ones(len(p), 1)], 1)
                pc = torch.bmm(ph[None], l2c[b].t()[None])[0][:, :3]              # transform_points
                depth = pc[:, 2].clone()
                K4 = torch.eye(4)
                K4[:3, :3] = K[b]
                uvw = torch.bmm(torch.cat([pc, torch.ones(len(pc), 1)], 1)[None], K4.t()[None])[0][:, :3]
                uv = (uvw / uvw[:, 2:3])[:, :2]
                g = uv.long()
                g = (image_scale * g.float()).long()
                m = (g[:, 0] > 0) & (g[:, 0] < W) & (g[:, 1] > 0) & (g[:, 1] < H) & (depth > depth_thres[cam])
                h_, w_ = img_feats[cam].shape[2:]
                gf = g.float()
                gf[:, 0] *= (w_ / W)
                gf[:, 1] *= (h_ / H)
                out.append((gf.long(), m, p))
            proj.append(out)
        for b in range(B):
            img = _t(img_feats[cam][b])
            h_, w_ = img.shape[1:]
            # ---- image-side gate
            pt_img = None
            for li in ifat_idx:
                ind, feat = levels[li]
                sel = ind[:, 0] == b
                g, m, p = proj[li][b]
                vf = torch.cat([_t(feat[sel])[m], p[m]], 1)
                canvas = torch.zeros(h_ + 1, w_ + 1, vf.shape[1])
                gy, gx = g[m][:, 1], g[m][:, 0]
                for j in range(len(vf)):                                          # sequential: last writer wins
                    canvas[gy[j], gx[j]] = vf[j]
                canvas = canvas[:-1, :-1].permute(2, 0, 1)[None]
                if li != ifat_idx[-1]:
                    canvas = F.conv2d(canvas, P("ifat.reduced_dim.%d.weight" % li), P("ifat.reduced_dim.%d.bias" % li))
                pt_img = canvas if pt_img is None else pt_img + canvas
            pt_img = F.conv2d(pt_img, P("ifat.reduced_dim2.weight"), P("ifat.reduced_dim2.bias"))
            gate = F.conv2d(img[None], P("ifat.reduced_dim3.weight"), P("ifat.reduced_dim3.bias"))
            att = torch.sigmoid(F.conv2d(gate + pt_img, P("ifat.spatial_basic.weight"), P("ifat.spatial_basic.bias"), padding=1))
            img = (img[None] * att)[0]
            img_gated[(b, ci)] = img
            ind, feat = levels[n_last]
            sel = ind[:, 0] == b
            g, m, p = proj[n_last][b]
            per[(b, ci)] = dict(grid=g[m], pts=p[m], feat=_t(feat[sel])[m], mask=m,
                                ifeat=img[:, g[m][:, 1], g[m][:, 0]].t())
    if debug is not None:
        debug.update(per=per)
    ncam = len(cams)
    max_ne = max(len(v["grid"]) for v in per.values())
    C = levels[n_last][1].shape[1]
    h_, w_ = next(iter(img_feats.values())).shape[2:]
    v_feat = np.zeros((B * ncam, max_ne, C), np.float32)
    v_i = np.zeros((B * ncam, max_ne, 256), np.float32)
    grid = np.zeros((B * ncam, max_ne, 2), np.float32)
    pts = np.zeros((B * ncam, max_ne, 3), np.float32)
    imgs = np.zeros((B * ncam, 256, h_, w_), np.float32)
    for (b, ci), v in per.items():
        n = len(v["grid"])
        i = b * ncam + ci
        v_feat[i, :n], v_i[i, :n], pts[i, :n] = v["feat"].numpy(), v["ifeat"].numpy(), v["pts"].numpy()
        grid[i, :n] = v["grid"].float().numpy()
        imgs[i] = img_gated[(b, ci)].numpy()
    grid = (_t(grid) / torch.tensor([w_, h_], dtype=torch.float32)).numpy()
    enh = actr_forward(sd, v_feat, grid, imgs, pts, v_i, prefix="pfat.")
    ind, feat = levels[n_last]
    out = np.array(feat, np.float32, copy=True)
    for b in range(B):
        rows = np.nonzero(ind[:, 0] == b)[0]
        for ci in range(ncam):
            m = per[(b, ci)]["mask"].numpy()
            out[rows[m]] += enh[b * ncam + ci][:int(m.sum())]
    return out


# ------------------------------------------------------------------------- TransFusion fusion layer (round 6)
def _undo_3d_augmentation(p, m):
    """`apply_3d_transformation(points, 'LIDAR', img_meta, reverse=True)` (TF/mmdet3d/models/fusion_layers/
    coord_transform.py:6-94 over core/points/base_points.py:77-141,199-205): the recorded flow undone in reverse order."""
    flow = list(m.get('transformation_3d_flow', []))
    if not flow:
        return p
    p = np.array(p, np.float32, copy=True)
    rot = np.asarray(m['pcd_rotation'], np.float32) if 'pcd_rotation' in m else np.eye(3, dtype=np.float32)
    trans = np.asarray(m['pcd_trans'], np.float32) if 'pcd_trans' in m else np.zeros(3, np.float32)
    scale = np.float32(m.get('pcd_scale_factor', 1.0))
    for op in flow[::-1]:
        if op == 'T':
            p = p + (-trans)
        elif op == 'S':
            p = p * np.float32(1.0 / scale)
        elif op == 'R':
            p = p @ np.linalg.inv(rot).astype(np.float32)
        elif op == 'HF':
            if m.get('pcd_horizontal_flip', False):
                p = p * np.array([1, -1, 1], np.float32)
        elif op == 'VF':
            if m.get('pcd_vertical_flip', False):
                p = p * np.array([-1, 1, 1], np.float32)
        else:
            raise ValueError(op)
    return p


def transfusion_project(pts, m):
    """`get_2d_coor_multi` + `projection` (TF/mmdet3d/models/fusion_layers/point_fusion.py:509-643) for one sample with the
    lidar -> camera chain of the nuScenes records composed into `m['lidar2cam']` [6, 4, 4] / `m['cam_intrinsic']` [6, 3, 3]:
    float64 numpy like the devkit arithmetic, cameras in order (a later camera overwrites), visibility on the ORIGINAL image,
    -> (coor_2d [n, 3] = (camera, x / w_pad, y / h_pad), coor_2d_o [n, 3] = (camera, x, y) in input-image pixels), zeros for
    points no camera sees."""
    p = _undo_3d_augmentation(np.asarray(pts, np.float32)[:, :3], m).astype(np.float64)
    n = len(p)
    c2, c2o = np.zeros((n, 3), np.float32), np.zeros((n, 3), np.float32)
    H, W = m['ori_shape'][:2]
    sf = np.asarray(m.get('scale_factor', [1.0, 1.0]), np.float32)[:2]
    off = np.float32(m.get('img_crop_offset', 0))
    hp, wp = m['input_shape'][:2]
    for idx in range(len(m['lidar2cam'])):
        T = np.asarray(m['lidar2cam'][idx], np.float64)
        K = np.asarray(m['cam_intrinsic'][idx], np.float64)
        pc = p @ T[:3, :3].T + T[:3, 3]
        depth = pc[:, 2]
        uvw = pc @ K.T
        with np.errstate(divide="ignore", invalid="ignore"):
            u, v = uvw[:, 0] / uvw[:, 2], uvw[:, 1] / uvw[:, 2]
        mask = (depth > 1.0) & (u > 1) & (u < W - 1) & (v > 1) & (v < H - 1)
        uv = np.stack([u[mask], v[mask]], 1).astype(np.float32)                 # points.new_tensor(...): fp32 from here on
        xy = uv * sf - off
        if m.get('flip', False):
            xy[:, 0] = np.float32(m['img_shape'][1]) - xy[:, 0]
        c2[mask, 0], c2o[mask, 0] = idx, idx
        c2o[mask, 1:] = xy
        c2[mask, 1:] = xy / np.array([wp, hp], np.float32)
    return c2, c2o


def transfusion_fusion(sd, pts_list, pts_feats, img, metas, prefix="actr.", num_cams=6, projection=None):
    """`point_fusion.ACTR.forward` with fusion_method 'sum' (point_fusion.py:342-394 split_param, :396-408 agg_param,
    :410-507 forward): per-sample projection, zero-padded per-(sample, camera) query lists in row order, the image feature of
    a query = level-0 map at (pixel // 4), ACTR on the padded lists, additive write-back.
    pts_list: per sample [n_b, 3] voxel centres (augmented frame); pts_feats [sum n_b, C]; img [B * 6, 256, h, w]."""
    B = len(pts_list)
    C, IC = pts_feats.shape[1], img.shape[1]
    # projection: per sample (coor_2d, coor_2d_o) to use instead of this composition's own -- the full-size parity tests hand
    # over the DEVICE's projection (after checking it against `transfusion_project`), so that a voxel whose pixel sits within
    # fp32 rounding of a boundary (camera border, multiple of 4) takes the same branch on both sides
    proj = projection if projection is not None else [transfusion_project(p, m) for p, m in zip(pts_list, metas)]
    lists = []
    for b in range(B):
        cam = proj[b][0][:, 0].astype(np.int64)
        lists += [np.nonzero(cam == n)[0] for n in range(num_cams)]
    max_pts = max(len(l) for l in lists)
    v_feat = np.zeros((B * num_cams, max_pts, C), np.float32)
    v_i = np.zeros((B * num_cams, max_pts, IC), np.float32)
    grid = np.zeros((B * num_cams, max_pts, 2), np.float32)
    qpts = np.zeros((B * num_cams, max_pts, 3), np.float32)
    starts = np.cumsum([0] + [len(p) for p in pts_list])
    for i, rows in enumerate(lists):
        b = i // num_cams
        k = len(rows)
        v_feat[i, :k] = pts_feats[starts[b]:starts[b + 1]][rows]
        grid[i, :k] = proj[b][0][rows, 1:3]
        qpts[i, :k] = np.asarray(pts_list[b], np.float32)[rows, :3]
        px = proj[b][1][rows, 1:].astype(np.int64) // 4                          # .to(torch.long) // 4
        v_i[i, :k] = img[i][:, px[:, 1], px[:, 0]].T
    enh = actr_forward(sd, v_feat, grid, img, qpts, v_i, prefix=prefix)
    out = np.array(pts_feats, np.float32, copy=True)
    for i, rows in enumerate(lists):
        b = i // num_cams
        out[starts[b] + rows] += enh[i, :len(rows)]
    return out


# ------------------------------------------------------------------------- Voxel-RCNN fusion glue (round 6)
def _kitti_lidar_to_img(p, l2i):
    """`Calibration.lidar_to_img` (VR/pcdet/utils/calibration_kitti.py:64-86) through the composed chain `l2i` [3, 4] whose
    third row is the rectified-camera depth row (make_golden.gen_vr_fusion composes it): float64 like the devkit's numpy."""
    ph = np.concatenate([p.astype(np.float64), np.ones((len(p), 1))], 1) @ np.asarray(l2i, np.float64).T
    return ph[:, :2] / ph[:, 2:3]


def _vr_voxel_points(ind_b, b, voxel_stride, aug):
    """Voxel corner (z, y, x) * stride * voxel_size + range minimum in fp32, then the augmentation records undone
    (spconv_backbone.py:671-705): scale, rotation about z, flip -- columns stay (z, y, x)."""
    vs = np.array([0.1, 0.05, 0.05], np.float32)
    lo = np.array([-3.0, -40.0, 0.0], np.float32)
    v = (ind_b[:, 1:] * voxel_stride).astype(np.float32) * vs + lo                 # (z, y, x)
    if aug is not None:
        if "noise_scale" in aug:
            v = v / np.float32(aug["noise_scale"][b])
        if "noise_rot" in aug:
            a = -np.float32(aug["noise_rot"][b])
            c, s_ = np.float32(np.cos(a)), np.float32(np.sin(a))
            xyz = v[:, ::-1]
            rot = np.array([[c, s_, 0], [-s_, c, 0], [0, 0, 1]], np.float32)          # common_utils.rotate_points_along_z
            v = (xyz @ rot)[:, ::-1]
        if "flip_x" in aug and aug["flip_x"][b]:
            v = v * np.array([1, -1, 1], np.float32)
        if "flip_y" in aug and aug["flip_y"][b]:
            v = v * np.array([1, 1, -1], np.float32)
    return np.ascontiguousarray(v, np.float32)


def _interp_bilinear(fmap, hw):
    import torch
    import torch.nn.functional as F
    return F.interpolate(_t(fmap), tuple(hw), mode="bilinear").numpy()


def voxel_rcnn_mvx(indices, features, fmap, lidar2img, image_hw, voxel_stride=1, aug=None, uv_rows=None):
    """`point_fusion(..., 'MVX')` with fuse_sum (spconv_backbone.py:650-760): the image feature at the truncated pixel of every
    voxel inside the image, bilinearly up-sampled map, added to the LiDAR row."""
    H, W = image_hw
    up = _interp_bilinear(fmap, (H, W))
    out = np.array(features, np.float32, copy=True)
    for b in range(fmap.shape[0]):
        rows = np.nonzero(indices[:, 0] == b)[0]
        v = _vr_voxel_points(indices[rows], b, voxel_stride, aug)
        # uv_rows [n, 2]: pixel coordinates to use instead of this composition's own (see transfusion_fusion: `projection`)
        uv = _kitti_lidar_to_img(v[:, ::-1], lidar2img[b]) if uv_rows is None else np.asarray(uv_rows)[rows]
        px = uv.astype(np.float32).astype(np.int64)                                 # torch.Tensor(voxels_2d).long()
        ok = (px[:, 1] >= 0) & (px[:, 1] < H) & (px[:, 0] >= 0) & (px[:, 0] < W)
        out[rows[ok]] += up[b][:, px[ok, 1], px[ok, 0]].T
    return out


def voxel_rcnn_pixels(indices, lidar2img, voxel_stride, aug=None):
    """Pixel coordinates [n, 2] (float64) of every voxel row: the projection of the two fusion points on its own."""
    uv = np.zeros((len(indices), 2))
    for b in np.unique(indices[:, 0]):
        rows = np.nonzero(indices[:, 0] == b)[0]
        uv[rows] = _kitti_lidar_to_img(_vr_voxel_points(indices[rows], int(b), voxel_stride, aug)[:, ::-1], lidar2img[int(b)])
    return uv


def voxel_rcnn_actr_fusion(sd, indices, features, fmap, lidar2img, image_hw, lt_cfg, voxel_stride=8, aug=None, num_layers=4,
                           prefix="", uv_rows=None):
    """`point_fusion(..., 'ACTRv2')` with fuse_sum (spconv_backbone.py:650-820): one zero-padded query list per sample (ALL its
    voxels, in row order), normalised un-truncated image coordinates, image features at the truncated pixel of the up-sampled
    map (zero outside the image), ACTRv2 = LocalTransformer + dual-query layer with the gate BEFORE the feed-forward blocks
    (VR/pcdet/models/model_utils/actr_transformer.py:496-512), added to the LiDAR rows."""
    import torch
    import torch.nn.functional as F
    H, W = image_hw
    B = fmap.shape[0]
    up = _interp_bilinear(fmap, (H, W))
    rows_b = [np.nonzero(indices[:, 0] == b)[0] for b in range(B)]
    n_max = max(len(r) for r in rows_b)
    C, IC = features.shape[1], fmap.shape[1]
    v_feat = np.zeros((B, n_max, C), np.float32)
    v_i = np.zeros((B, n_max, IC), np.float32)
    grid = np.zeros((B, n_max, 2), np.float32)
    qpts = np.zeros((B, n_max, 3), np.float32)
    for b, rows in enumerate(rows_b):
        v = _vr_voxel_points(indices[rows], b, voxel_stride, aug)
        uv = _kitti_lidar_to_img(v[:, ::-1], lidar2img[b]) if uv_rows is None else np.asarray(uv_rows, np.float64)[rows]
        px = uv.astype(np.float32).astype(np.int64)
        ok = (px[:, 1] >= 0) & (px[:, 1] < H) & (px[:, 0] >= 0) & (px[:, 0] < W)
        k = len(rows)
        v_feat[b, :k] = features[rows]
        v_i[b, :k][ok] = up[b][:, px[ok, 1], px[ok, 0]].T
        grid[b, :k] = (uv / np.array([W, H])).astype(np.float32)
        qpts[b, :k] = v[:, ::-1]                                                    # pts_b[..., inv_idx] = (x, y, z)
    enh = actr_v2_forward(sd, v_feat, grid, fmap, qpts, v_i, lt_cfg, num_layers=num_layers, prefix=prefix)
    out = np.array(features, np.float32, copy=True)
    for b, rows in enumerate(rows_b):
        out[rows] += enh[b, :len(rows)]
    return out


def actr_v2_forward(sd, v_feat, grid, i_feat, lidar_grid, v_i_feat, lt_cfg, num_layers=4, n_heads=8, n_points=4, prefix=""):
    """ACTRv2 of the Voxel-RCNN tree (VR/pcdet/models/model_utils/actr.py:131-187, actr_transformer.py:473-512): as
    `actr_forward`, with a LocalTransformer over the LiDAR queries in front of every dual-query layer (:496-498) and the
    bidirectional gate right after the attention, the two feed-forward blocks on the gated streams (:503-512)."""
    import torch
    import torch.nn.functional as F
    P = lambda k: _t(sd[prefix + k])
    v_feat, grid, i_feat, lidar_grid, v_i_feat = [_t(np.asarray(a, np.float32)) for a in
                                                  (v_feat, grid, i_feat, lidar_grid, v_i_feat)]
    N, Q, C = v_feat.shape
    H, W = i_feat.shape[2:]
    qi = F.conv1d(v_i_feat.transpose(1, 2), P("i_input_proj.0.weight"), P("i_input_proj.0.bias"))
    qi = F.group_norm(qi, 32, P("i_input_proj.1.weight"), P("i_input_proj.1.bias"), 1e-5).transpose(1, 2)
    d = lidar_grid[..., 0] / 60.0 * (2 * np.pi)
    dim_t = torch.arange(C, dtype=torch.float32)
    dim_t = 10000 ** (2 * torch.div(dim_t, 2, rounding_mode="floor") / C)
    pd = d[:, :, None] / dim_t
    q_pos = torch.stack((pd[:, :, 0::2].sin(), pd[:, :, 1::2].cos()), dim=3).flatten(2)
    src = F.group_norm(F.conv2d(i_feat, P("input_proj.0.0.weight"), P("input_proj.0.0.bias")), 32,
                       P("input_proj.0.1.weight"), P("input_proj.0.1.bias"), 1e-5)
    src = src.flatten(2).transpose(1, 2)
    q = v_feat
    D = C // n_heads
    lin = lambda x, k: F.linear(x, P(k + ".weight"), P(k + ".bias"))                 # noqa: E731
    for i in range(num_layers):
        lt_sd = {k[len(prefix + "transformer.encoder.lidar_attns.%d." % i):]: v for k, v in sd.items()
                 if k.startswith(prefix + "transformer.encoder.lidar_attns.%d." % i)}
        q = _t(local_transformer(lt_sd, lidar_grid.numpy(), q.permute(0, 2, 1).contiguous().numpy(), lt_cfg["npoint"],
                                 lt_cfg["radius"], lt_cfg["nsample"], num_layers=lt_cfg["num_layers"]))
        L = "transformer.encoder.layers.%d." % i
        value = lin(src, L + "self_attn.value_proj")
        query, iq = q + q_pos, qi + q_pos
        off = lin(query, L + "self_attn.sampling_offsets")
        aw = lin(query + iq, L + "self_attn.attention_weights")
        aw = torch.softmax(aw.view(N, Q, n_heads, n_points), -1).view(N, Q, n_heads, 1, n_points)
        off = off.view(N, Q, n_heads, 1, n_points, 2)
        loc = grid[:, :, None, None, None, :] + off / torch.tensor([W, H], dtype=torch.float32)
        att = orc.ms_deform_attn(value.view(N, H * W, n_heads, D).numpy(), [(H, W)], loc.numpy(), aw.numpy())
        qi = F.layer_norm(qi + lin(_t(att), L + "self_attn.output_proj"), (C,), P(L + "norm1.weight"), P(L + "norm1.bias"))
        fuse = (q + qi).transpose(1, 2)
        s1 = torch.sigmoid(F.conv1d(fuse, P(L + "fusion_layer.b_conv1d.weight"), P(L + "fusion_layer.b_conv1d.bias"))).transpose(1, 2)
        s2 = torch.sigmoid(F.conv1d(fuse, P(L + "fusion_layer.a_conv1d.weight"), P(L + "fusion_layer.a_conv1d.bias"))).transpose(1, 2)
        q, qi = q + qi * s1, qi + q * s2
        qi = F.layer_norm(qi + lin(F.relu(lin(qi, L + "linear1")), L + "linear2"), (C,), P(L + "norm2.weight"), P(L + "norm2.bias"))
        q = F.layer_norm(q + lin(F.relu(lin(q, L + "linear3")), L + "linear4"), (C,), P(L + "norm3.weight"), P(L + "norm3.bias"))
    return q.numpy()


# ------------------------------------------------------------------------- TransFusion head (section 8f row 3)
def transfusion_head(sd, x, num_proposals, num_classes=10, num_heads=8, nms_kernel_size=3, dataset="nuScenes",
                     num_decoder_layers=1, heads=("center", "height", "dim", "rot", "vel", "heatmap"), bn_eps=1e-5):
    """TransFusionHead.forward_single, LiDAR-only branch (fuse_img=False, initialize_by_heatmap=True), eval mode
    (TF/mmdet3d/models/dense_heads/transfusion_head.py:797-1030; decoder layer :82-122, attention :255-505,
    position embedding :30-41, prediction FFN :507-591).  torch-CPU fp32, one statement per reference statement.
    Returns (dict of numpy arrays with the reference's keys, query_labels)."""
    import torch
    import torch.nn.functional as F
    P = lambda k: _t(sd[k])
    x = _t(np.asarray(x, np.float32))
    B, _, H, W = x.shape

    def bn(prefix, v):
        return F.batch_norm(v, P(prefix + ".running_mean"), P(prefix + ".running_var"), P(prefix + ".weight"),
                            P(prefix + ".bias"), False, 0.0, bn_eps)

    def posembed(prefix, xy):                                   # :38-41 on [B, P, 2]
        v = F.conv1d(xy.transpose(1, 2), P(prefix + ".position_embedding_head.0.weight"),
                     P(prefix + ".position_embedding_head.0.bias"))
        v = F.relu(bn(prefix + ".position_embedding_head.1", v))
        return F.conv1d(v, P(prefix + ".position_embedding_head.3.weight"), P(prefix + ".position_embedding_head.3.bias"))

    def mha(prefix, q, k, v):                                   # :255-505, [L, N, E] operands, no masks, eval
        E = q.shape[-1]
        w, b = P(prefix + ".in_proj_weight"), P(prefix + ".in_proj_bias")
        hd = E // num_heads
        qq = F.linear(q, w[:E], b[:E]) * (float(hd) ** -0.5)
        kk = F.linear(k, w[E:2 * E], b[E:2 * E])
        vv = F.linear(v, w[2 * E:], b[2 * E:])
        L, N = q.shape[0], q.shape[1]
        qq = qq.contiguous().view(L, N * num_heads, hd).transpose(0, 1)
        kk = kk.contiguous().view(-1, N * num_heads, hd).transpose(0, 1)
        vv = vv.contiguous().view(-1, N * num_heads, hd).transpose(0, 1)
        a = torch.softmax(torch.bmm(qq, kk.transpose(1, 2)), -1)
        o = torch.bmm(a, vv).transpose(0, 1).contiguous().view(L, N, E)
        return F.linear(o, P(prefix + ".out_proj.weight"), P(prefix + ".out_proj.bias"))

    lidar_feat = F.conv2d(x, P("shared_conv.weight"), P("shared_conv.bias"), padding=1)                  # :808
    flat = lidar_feat.view(B, lidar_feat.shape[1], -1)
    ys, xs = torch.meshgrid(torch.linspace(0, H - 1, H), torch.linspace(0, W - 1, W), indexing="ij")    # :758-765
    bev_pos = torch.stack([xs + 0.5, ys + 0.5], 0).view(1, 2, -1).permute(0, 2, 1).repeat(B, 1, 1)
    hmid = F.relu(bn("heatmap_head.0.bn", F.conv2d(lidar_feat, P("heatmap_head.0.conv.weight"), None, padding=1)))
    dense_heatmap = F.conv2d(hmid, P("heatmap_head.1.weight"), P("heatmap_head.1.bias"), padding=1)      # :843
    heatmap = dense_heatmap.sigmoid()
    pad = nms_kernel_size // 2
    local_max = torch.zeros_like(heatmap)
    inner = F.max_pool2d(heatmap, kernel_size=nms_kernel_size, stride=1, padding=0)
    local_max[:, :, pad:H - pad, pad:W - pad] = inner                                                    # :851-854
    exempt = {"nuScenes": (8, 9), "Waymo": (1, 2)}.get(dataset, ())
    for c in exempt:                                                                                     # :856-861
        local_max[:, c] = heatmap[:, c]
    heatmap = (heatmap * (heatmap == local_max)).view(B, num_classes, -1)
    top = heatmap.view(B, -1).argsort(dim=-1, descending=True, stable=True)[..., :num_proposals]          # :866
    top_class = top // heatmap.shape[-1]
    top_index = top % heatmap.shape[-1]
    query_feat = flat.gather(index=top_index[:, None, :].expand(-1, flat.shape[1], -1), dim=-1)
    one_hot = F.one_hot(top_class, num_classes=num_classes).permute(0, 2, 1)
    query_feat = query_feat + F.conv1d(one_hot.float(), P("class_encoding.weight"), P("class_encoding.bias"))
    query_pos = bev_pos.gather(index=top_index[:, :, None].expand(-1, -1, 2), dim=1)                     # :878
    rets = []
    for i in range(num_decoder_layers):
        D = "decoder.%d" % i
        qpe = posembed(D + ".self_posembed", query_pos).permute(2, 0, 1)                                  # :92-99
        kpe = posembed(D + ".cross_posembed", bev_pos).permute(2, 0, 1)
        q = query_feat.permute(2, 0, 1)
        k = flat.permute(2, 0, 1)
        q = F.layer_norm(q + mha(D + ".self_attn", q + qpe, q + qpe, q + qpe), (q.shape[-1],),
                         P(D + ".norm1.weight"), P(D + ".norm1.bias"))                                    # :104-108
        q = F.layer_norm(q + mha(D + ".multihead_attn", q + qpe, k + kpe, k + kpe), (q.shape[-1],),
                         P(D + ".norm2.weight"), P(D + ".norm2.bias"))                                    # :110-114
        f = F.linear(F.relu(F.linear(q, P(D + ".linear1.weight"), P(D + ".linear1.bias"))),
                     P(D + ".linear2.weight"), P(D + ".linear2.bias"))
        q = F.layer_norm(q + f, (q.shape[-1],), P(D + ".norm3.weight"), P(D + ".norm3.bias"))            # :116-118
        query_feat = q.permute(1, 2, 0)
        res = {}
        for h in heads:                                                                                   # :585-589
            pre = "prediction_heads.%d.%s" % (i, h)
            v = F.relu(bn(pre + ".0.bn", F.conv1d(query_feat, P(pre + ".0.conv.weight"), None)))
            res[h] = F.conv1d(v, P(pre + ".1.weight"), P(pre + ".1.bias"))
        res["center"] = res["center"] + query_pos.permute(0, 2, 1)                                        # :899
        rets.append(res)
        query_pos = res["center"].clone().permute(0, 2, 1)
    out = {k: torch.cat([r[k] for r in rets], -1) for k in rets[0]}                                       # :1022-1028
    out["query_heatmap_score"] = heatmap.gather(index=top_index[:, None, :].expand(-1, num_classes, -1), dim=-1)
    out["dense_heatmap"] = dense_heatmap
    return {k: v.numpy() for k, v in out.items()}, top_class.numpy()


def transfusion_get_bboxes(preds, query_labels, num_proposals, coder, num_classes=10):
    """TransFusionHead.get_bboxes with nms_type=None (transfusion_head.py:1285-1312,1326-1376) over
    TransFusionBBoxCoder.decode(filter=True) (core/bbox/coders/transfusion_bbox_coder.py:41-128), per sample.
    Returns a list of (boxes [n, 7 or 9], scores [n], labels [n])."""
    K = num_proposals
    score = 1.0 / (1.0 + np.exp(-preds["heatmap"][..., -K:].astype(np.float32)))
    one_hot = np.eye(num_classes, dtype=np.float32)[query_labels].transpose(0, 2, 1)
    score = (score * preds["query_heatmap_score"] * one_hot).astype(np.float32)
    labels, scores = score.argmax(1), score.max(1)
    center = preds["center"][..., -K:].astype(np.float32).copy()
    f32 = np.float32
    center[:, 0] = center[:, 0] * f32(coder["out_size_factor"]) * f32(coder["voxel_size"][0]) + f32(coder["pc_range"][0])
    center[:, 1] = center[:, 1] * f32(coder["out_size_factor"]) * f32(coder["voxel_size"][1]) + f32(coder["pc_range"][1])
    dim = np.exp(preds["dim"][..., -K:].astype(np.float32))
    height = preds["height"][..., -K:] - dim[:, 2:3] * f32(0.5)
    rot = np.arctan2(preds["rot"][..., -K:][:, 0:1], preds["rot"][..., -K:][:, 1:2])
    parts = [center, height, dim, rot] + ([preds["vel"][..., -K:]] if "vel" in preds else [])
    boxes = np.concatenate(parts, 1).transpose(0, 2, 1).astype(np.float32)
    rng = np.asarray(coder["post_center_range"], np.float32)
    mask = (boxes[..., :3] >= rng[:3]).all(2) & (boxes[..., :3] <= rng[3:]).all(2)
    if coder.get("score_threshold"):
        mask &= scores > coder["score_threshold"]
    return [(boxes[b][mask[b]], scores[b][mask[b]], labels[b][mask[b]]) for b in range(len(boxes))]
