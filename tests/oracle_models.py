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


def _conv(sd, prefix, x, ks, stride, padding, subm, key=None):
    w = sd[prefix + ".weight"]
    rb = x.rulebooks.get(key) if key is not None else None
    if rb is None:
        rb = orc.get_indice_pairs(x.indices, x.batch, x.shape, ks, stride, padding, [1, 1, 1], subm)
        if key is not None:
            x.rulebooks[key] = rb
    outids, pairs, num, oshape = rb
    y = orc.indice_conv(x.features, w, pairs, num, len(outids), subm)
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


def voxel_backbone8x(sd, voxel_features, coors, batch_size, sparse_shape):
    """VR/pcdet/models/backbones_3d/spconv_backbone.py:135-243 (LiDAR branch)."""
    x = SpTensor(np.asarray(voxel_features, np.float32), np.asarray(coors, np.int32), sparse_shape, batch_size)
    x = _cbr(sd, "conv_input.0", "conv_input.1", x, [3, 3, 3], [1, 1, 1], [1, 1, 1], 1, "subm1")
    c1 = _cbr(sd, "conv1.0.0", "conv1.0.1", x, [3, 3, 3], [1, 1, 1], [1, 1, 1], 1, "subm1")
    outs = {"x_conv1": c1}
    cur = c1
    for s, pad in ((2, [1, 1, 1]), (3, [1, 1, 1]), (4, [0, 1, 1])):
        cur = _cbr(sd, "conv%d.0.0" % s, "conv%d.0.1" % s, cur, [3, 3, 3], [2, 2, 2], pad, 0)
        for j in (1, 2):
            cur = _cbr(sd, "conv%d.%d.0" % (s, j), "conv%d.%d.1" % (s, j), cur, [3, 3, 3], [1, 1, 1], [1, 1, 1], 1,
                       "subm%d" % s)
        outs["x_conv%d" % s] = cur
    out = _cbr(sd, "conv_out.0", "conv_out.1", cur, [3, 1, 1], [2, 1, 1], [0, 0, 0], 0)
    return out, outs
