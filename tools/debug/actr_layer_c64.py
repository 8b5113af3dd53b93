#!/usr/bin/env python3
"""Fused dual-query layer (csrc/actr.hip, ffn.hip) vs its unfused torch composition at d_model 64 (ACTRv2 / Voxel-RCNN)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]
import torch
import torch.nn.functional as F
from dualfusion import ops
from dualfusion.actr import DeformableTransformerFusionEncoderLayer

dev = torch.device("cuda:0")
for C in (64, 128):
    torch.manual_seed(0)
    L = DeformableTransformerFusionEncoderLayer(d_model=C, q_model=C, d_ffn=1024, n_levels=1, n_heads=8, n_points=4,
                                                hybrid_cfg=dict(attn_layer='BiGateSum1D_2', q_method='sum', q_rep_place=['weight'])).to(dev).eval()
    with torch.no_grad():
        for p in L.parameters():
            if p.dim() > 1:
                p.normal_(0, 0.05)
            else:
                p.normal_(0, 0.1)
        L.norm1.weight.add_(1); L.norm2.weight.add_(1); L.norm3.weight.add_(1)
    N, Q, H, W = 2, 260, 24, 80
    src = torch.randn(N, H * W, C, device=dev)
    ref = torch.rand(N, Q, 1, 2, device=dev)
    shp = torch.tensor([[H, W]], device=dev)
    lsi = torch.zeros(1, dtype=torch.long, device=dev)
    q, qi, qp = torch.randn(N, Q, C, device=dev), torch.randn(N, Q, C, device=dev), torch.randn(N, Q, C, device=dev)
    with torch.no_grad():
        a_q, a_qi = L(src, None, ref, shp, lsi, None, q_pos=qp, q_feat=q, q_i_feat=qi)
    with torch.enable_grad():
        b_q, b_qi = L(src, None, ref, shp, lsi, None, q_pos=qp, q_feat=q, q_i_feat=qi)
    print("C=%d fused vs unfused: q %.3e  qi %.3e (rel to max)" % (C, float((a_q - b_q).abs().max() / b_q.abs().max()),
                                                                 float((a_qi - b_qi).abs().max() / b_qi.abs().max())))
    # pieces
    with torch.no_grad():
        A, Bw = ops.actr_prep(q.contiguous(), qi.contiguous(), qp.contiguous())
        print("  actr_prep: A %.2e  Bw %.2e" % (float((A - (q + qp)).abs().max()), float((Bw - (q + qp + qi + qp)).abs().max())))
        x = torch.randn(N, Q, C, device=dev); y = torch.randn(N, Q, C, device=dev)
        ln = ops.add_layernorm(x, y, L.norm1.weight, L.norm1.bias, L.norm1.eps)
        print("  add_layernorm %.2e" % float((ln - L.norm1(x + y)).abs().max()))
        f1 = L._ffn(x, L.linear1, L.linear2, L.norm2, "_t1")
        f1r = L.norm2(x + L.linear2(F.relu(L.linear1(x))))
        print("  ffn (supported=%s) %.2e" % (ops.ffn_supported(C, 1024), float((f1 - f1r).abs().max())))
        fa, fb = L._ffn_pair(x, y)
        print("  ffn_pair %.2e %.2e" % (float((fa - f1r).abs().max()), float((fb - L.norm3(y + L.linear4(F.relu(L.linear3(y))))).abs().max())))
        g = L.fusion_layer
        bs = ops.bigate_sum(x, y, g.b_conv1d.weight.view(-1), g.b_conv1d.bias, g.a_conv1d.weight.view(-1), g.a_conv1d.bias)
        r1, r2 = g(x, y)
        print("  bigate %.2e %.2e" % (float((bs[0] - r1).abs().max()), float((bs[1] - r2).abs().max())))
        sa = L.self_attn
        value = sa.project_value(src)
        out = ops.ms_deform_attn_fused(value, shp, lsi, ref[:, :, 0, :].contiguous(), sa.sampling_offsets(A), sa.attention_weights(Bw),
                                       sa.n_levels, sa.n_points, None, None)
        outr = sa(q + qp, ref, src, shp, lsi, None, i_query=qi + qp)
        print("  msda fused (+output_proj) %.2e" % float((sa.output_proj(out) - outr).abs().max()))
