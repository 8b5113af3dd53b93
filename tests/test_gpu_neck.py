"""BEV neck on the sparse-convolution kernels: the RPN module's row path against the
torch fp32 CPU composition of the same module (Conv2d / ConvTranspose2d / BatchNorm2d / ReLU), and the channels-last
dense() against SparseConvTensor.dense()."""
import numpy as np
import pytest

import detgen

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _det_module(m):
    sd = detgen.det_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return m.eval()


@pytest.mark.parametrize("B,H,W", [(2, 44, 36), (1, 180, 180)])
def test_rpn_rows_vs_torch_cpu(B, H, W):
    from dualfusion.necks import RPN
    dev = torch.device("cuda:0")
    m = _det_module(RPN([5, 5], [1, 2], [128, 256], [1, 2], [256, 256], 256))
    x = torch.from_numpy(detgen.randn("neck_x_%d_%d" % (H, W), (B, 256, H, W)))
    x = x * (torch.rand(B, 1, H, W, generator=torch.Generator().manual_seed(1)) < 0.3)     # BEV maps are mostly empty
    with torch.no_grad():
        ref = m.forward_reference(x)                       # torch CPU fp32 = the oracle for this floating-point row
        md = m.to(dev)
        y = md(x.to(dev))
        y_lib = md.forward_reference(x.to(dev))            # MIOpen / hipBLASLt composition, for reference only
    assert tuple(y.shape) == tuple(ref.shape) == (B, 512, H, W)
    err = float((y.cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err                                 # fp32-grade (north_star tolerance 1e-3): fp32 (torch CPU) against fp32 grade
    assert float((y_lib.cpu() - ref).abs().max() / ref.abs().max()) < 1e-3


def test_rpn_concatenation_written_in_place(monkeypatch):
    """Round 3: the last layer of each upsampling stack writes its columns of the concatenated rows (fp32 and operand
    split) itself.  Same bits as the separate maps + torch.cat, and the split rows handed to the head are the split of
    the fp32 rows."""
    from dualfusion import ops
    from dualfusion.necks import RPN, _rows_of
    dev = torch.device("cuda:0")
    m = _det_module(RPN([5, 5], [1, 2], [128, 256], [1, 2], [256, 256], 256)).to(dev)
    x = torch.from_numpy(detgen.randn("neck_cat", (2, 256, 44, 36))).to(dev)
    with torch.no_grad():
        y = m(x)
        rows, split = _rows_of(y)
        assert split is not None and tuple(rows.shape) == (2 * 44 * 36, 512)
        assert torch.equal(split, ops.split_rows(rows))
        monkeypatch.setenv("DF3D_NECK_CAT", "0")
        m.__dict__.pop("_row_plan", None)
        y0 = m(x)
        assert _rows_of(y0)[1] is None
    assert torch.equal(y, y0)


def test_rpn_parameter_layout_and_exact_path():
    """state_dict keys of the reference checkpoint layout; DF3D_CONV_PRECISION=fp32 path through the same module."""
    from dualfusion import ops
    from dualfusion.necks import RPN
    dev = torch.device("cuda:0")
    m = RPN([5, 5], [1, 2], [128, 256], [1, 2], [256, 256], 256)
    keys = set(m.state_dict())
    assert {"blocks.0.1.weight", "blocks.0.2.running_var", "blocks.0.4.weight", "blocks.1.16.weight",
            "deblocks.0.0.weight", "deblocks.1.0.weight", "deblocks.1.1.bias"} <= keys
    assert tuple(m.state_dict()["deblocks.1.0.weight"].shape) == (256, 256, 2, 2)
    m = _det_module(m)
    x = torch.from_numpy(detgen.randn("neck_exact", (1, 256, 20, 24)))
    with torch.no_grad():
        ref = m.forward_reference(x)
        old = ops.CONV_PRECISION
        ops.CONV_PRECISION = "fp32"
        try:
            y = m.to(dev)(x.to(dev))
        finally:
            ops.CONV_PRECISION = old
    assert float((y.cpu() - ref).abs().max() / ref.abs().max()) < 1e-4


def test_dense_rows_equals_dense_view():
    from dualfusion import ops
    dev = torch.device("cuda:0")
    shape, batch, C = [2, 30, 28], 3, 128
    ind = detgen.clustered_voxels("dnr", batch, shape, n_seeds=5, walk=150)
    f = detgen.randn("dnrf", (len(ind), C))
    ft, it = torch.from_numpy(f).to(dev), torch.from_numpy(ind).to(dev)
    rows = ops.sparse_to_dense_rows(ft, it, batch, shape)
    dense = ops.sparse_to_dense(ft, it, batch, shape)                      # [B, C, D, H, W]
    want = dense.view(batch, C * shape[0], shape[1], shape[2]).permute(0, 2, 3, 1).reshape(-1, C * shape[0])
    assert torch.equal(rows, want)


def test_pipeline_with_neck_equals_dense_then_neck():
    """CenterPointHotPath(neck=RPN): backbone hands over pixel rows (no NCHW volume); same values as dense() -> RPN."""
    from dualfusion import synth
    from dualfusion.necks import RPN
    from dualfusion.pipeline import CenterPointHotPath
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    neck = _det_module(RPN([5, 5], [1, 2], [128, 256], [1, 2], [256, 256], 256)).to(dev)
    model = CenterPointHotPath().eval().to(dev)
    pts = [torch.from_numpy(synth.nusc_sweep(seed=s)).to(dev) for s in (0, 1)]
    with torch.no_grad():
        bev, _ = model(pts)
        want = neck(bev)
        want_ref = neck.forward_reference(bev)
        with_neck = CenterPointHotPath(neck=neck).eval().to(dev)
        with_neck.backbone.load_state_dict(model.backbone.state_dict())
        got, multi = with_neck(pts)
    assert tuple(got.shape) == (2, 512, 180, 180) and set(multi) == {"conv1", "conv2", "conv3", "conv4"}
    assert torch.equal(got, want)
    assert float((got - want_ref).abs().max() / want_ref.abs().max()) < 1e-3


def test_second_fpn_rows_vs_torch_cpu():
    """TransFusion tree: SECOND -> SECONDFPN (transfusion_nusc_voxel_LC.py:168-183) on the row kernels."""
    from dualfusion.necks import SECOND, SECONDFPN
    from dualfusion.registry import MM_BACKBONES, MM_NECKS, build_from_cfg
    dev = torch.device("cuda:0")
    bb = build_from_cfg(dict(type="SECOND", in_channels=256, out_channels=[128, 256], layer_nums=[5, 5],
                             layer_strides=[1, 2], norm_cfg=dict(type="BN", eps=0.001, momentum=0.01),
                             conv_cfg=dict(type="Conv2d", bias=False)), MM_BACKBONES)
    fpn = build_from_cfg(dict(type="SECONDFPN", in_channels=[128, 256], out_channels=[256, 256],
                              upsample_strides=[1, 2], norm_cfg=dict(type="BN", eps=0.001, momentum=0.01),
                              upsample_cfg=dict(type="deconv", bias=False), use_conv_for_no_stride=True), MM_NECKS)
    assert isinstance(bb, SECOND) and isinstance(fpn, SECONDFPN)
    assert {"blocks.0.0.weight", "blocks.1.15.weight", "blocks.1.16.running_mean"} <= set(bb.state_dict())
    assert tuple(fpn.state_dict()["deblocks.0.0.weight"].shape) == (256, 128, 1, 1)      # conv for stride 1
    assert tuple(fpn.state_dict()["deblocks.1.0.weight"].shape) == (256, 256, 2, 2)      # deconv [cin, cout, s, s]
    bb, fpn = _det_module(bb), _det_module(fpn)
    x = torch.from_numpy(detgen.randn("second_x", (2, 256, 40, 52)))
    x = x * (torch.rand(2, 1, 40, 52, generator=torch.Generator().manual_seed(2)) < 0.3)
    with torch.no_grad():
        ref_ms = bb.forward_reference(x)
        ref = fpn.forward_reference(ref_ms)[0]
        bb, fpn = bb.to(dev), fpn.to(dev)
        ms = bb(x.to(dev))
        out = fpn(ms)[0]
        assert getattr(ms[1], "_df3d_rows", None) is not None          # stayed on rows between the two modules
        # default deconv for stride 1 (use_conv_for_no_stride=False) is a 1x1 ConvTranspose2d
        fpn2 = _det_module(SECONDFPN(in_channels=[128, 256], out_channels=[256, 256], upsample_strides=[1, 2]))
        ref2 = fpn2.forward_reference(ref_ms)[0]
        out2 = fpn2.to(dev)(ms)[0]
    for a, b in zip(ms, ref_ms):
        assert tuple(a.shape) == tuple(b.shape)
        assert float((a.cpu() - b).abs().max() / b.abs().max()) < 1e-3
    assert tuple(out.shape) == tuple(ref.shape) == (2, 512, 40, 52)
    assert float((out.cpu() - ref).abs().max() / ref.abs().max()) < 1e-3
    assert float((out2.cpu() - ref2).abs().max() / ref2.abs().max()) < 1e-3


def test_rpn_bf16_mode():
    """DF3D_CONV_PRECISION=bf16: the neck's 13 layers on the bf16 kernel, within bf16 rounding of the fp32 module."""
    from dualfusion import ops
    from dualfusion.necks import RPN
    dev = torch.device("cuda:0")
    m = _det_module(RPN([5, 5], [1, 2], [128, 256], [1, 2], [256, 256], 256))
    x = torch.from_numpy(detgen.randn("neck_bf16", (1, 256, 40, 48)))
    with torch.no_grad():
        ref = m.forward_reference(x)
        md = m.to(dev)
        old = ops.CONV_PRECISION
        ops.CONV_PRECISION = "bf16"
        try:
            y = md(x.to(dev))
        finally:
            ops.CONV_PRECISION = old
            md.train(False)                       # drops the plan that holds the bf16 filter banks
    assert float((y.cpu() - ref).abs().max() / ref.abs().max()) < 5e-2
    assert float((y.cpu() - ref).abs().mean() / ref.abs().mean()) < 1e-2


def _train_case(mod_cpu, run, x, grad_l2=1e-3, flip_l2=None):
    """Train-mode forward + backward of `run(module, x)` on the CPU in float64 (the oracle) and on the device through the
    row kernels: outputs, input gradient, every parameter gradient and the running statistics."""
    import copy
    flip_l2 = grad_l2 if flip_l2 is None else flip_l2
    ref = copy.deepcopy(mod_cpu).double().train()
    xr = x.double().requires_grad_(True)
    yr = run(ref, xr)
    g = torch.from_numpy(detgen.randn("neck_train_g", tuple(yr.shape))).double()
    (yr * g).sum().backward()
    dev = torch.device("cuda:0")
    md = copy.deepcopy(mod_cpu).to(dev).train()
    xd = x.to(dev).requires_grad_(True)
    yd = run(md, xd)
    (yd * g.float().to(dev)).sum().backward()
    rel = lambda a, b: float((a.detach().cpu().double() - b).abs().max() / max(1e-12, float(b.abs().max())))
    # gradients: a pre-activation within rounding of zero flips its ReLU mask, which moves single gradient entries by more
    # than rounding -- the L2 distance is the stable measure, the largest entry gets a looser bound
    l2 = lambda a, b: float((a.detach().cpu().double() - b).norm() / max(1e-12, float(b.norm())))
    assert rel(yd, yr.detach()) < 1e-3
    assert l2(xd.grad, xr.grad) < grad_l2 and rel(xd.grad, xr.grad) < 10 * grad_l2
    pr = dict(ref.named_parameters())
    errs = []
    for name, p in md.named_parameters():
        assert p.grad is not None, name
        if float(pr[name].grad.norm()) < 1e-9:             # identically zero (a conv bias in front of a BatchNorm)
            assert float(p.grad.abs().max()) < 1e-3, name
            continue
        errs.append(l2(p.grad, pr[name].grad))
        assert errs[-1] < flip_l2 and rel(p.grad, pr[name].grad) < 10 * flip_l2, (name, errs[-1])
    assert float(np.median(errs)) < 1e-3, errs
    br = dict(ref.named_buffers())
    for name, b in md.named_buffers():
        if b.dtype.is_floating_point:
            assert rel(b, br[name]) < 1e-3, name
        else:
            assert int(b) == int(br[name]), name


def test_rpn_training_on_row_kernels_vs_float64():
    """RPN in train mode (conv -> BatchNorm with batch statistics -> ReLU, strided conv, transposed conv, 1x1 conv):
    forward, input / filter / BN gradients and running statistics against the torch composition in float64."""
    from dualfusion.necks import RPN
    m = _det_module(RPN([1, 2], [1, 2], [32, 64], [1, 2], [32, 48], 16))
    x = torch.from_numpy(detgen.randn("neck_train_x", (2, 16, 20, 28)))
    x = x * (torch.rand(2, 1, 20, 28, generator=torch.Generator().manual_seed(3)) < 0.5)
    _train_case(m, lambda mod, t: mod(t) if t.is_cuda else mod.forward_reference(t), x)


def test_second_fpn_training_on_row_kernels_vs_float64():
    from dualfusion.necks import SECOND, SECONDFPN
    from torch import nn

    class Both(nn.Module):
        def __init__(self):
            super(Both, self).__init__()
            self.bb = SECOND(in_channels=16, out_channels=[32, 64], layer_nums=[1, 1], layer_strides=[1, 2])
            self.fpn = SECONDFPN(in_channels=[32, 64], out_channels=[32, 32], upsample_strides=[1, 2])

    m = _det_module(Both())
    x = torch.from_numpy(detgen.randn("second_train_x", (2, 16, 16, 24)))

    def run(mod, t):
        if t.is_cuda:
            return mod.fpn(mod.bb(t))[0]
        return mod.fpn.forward_reference(mod.bb.forward_reference(t))[0]
    _train_case(m, run, x)


def test_round4_tail_fusions_are_bit_identical_to_the_separate_passes():
    """The small fusions around the camera adapter and the dense map (round 4) against the passes they replace:
    dense rows written as split rows == df3d_split_rows(dense rows); write-back with 16-byte accesses + split rows ==
    df3d_fusion_writeback (+ df3d_split_rows); gate finish with the summary's bias added inside == the element-wise add in
    front of df3d_gate_finish; gate rows on a given winner map == df3d_gate_scatter_rows."""
    import ctypes
    from dualfusion import _lib, ops
    from dualfusion import spconv
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    P = lambda t: ctypes.c_void_p(t.data_ptr())      # noqa: E731
    # ---- dense rows
    B, D, H, W, C, n = 2, 2, 45, 37, 128, 1500
    flat = torch.randperm(B * D * H * W, generator=g)[:n].sort().values
    ind = torch.stack([flat // (D * H * W), (flat // (H * W)) % D, (flat // W) % H, flat % W], 1).to(torch.int32).to(dev)
    feats = torch.randn(n, C, generator=g).to(dev)
    x = spconv.SparseConvTensor(feats, ind, [D, H, W], B)
    rows = x.dense_rows()
    assert torch.equal(x.dense_rows(split=True), ops.split_rows(rows))
    # ---- write-back
    ncam, max_ne, nq = 6, 300, 1200
    n4 = 1000
    ind4 = torch.stack([torch.arange(n4) // 500, torch.zeros(n4, dtype=torch.long), torch.arange(n4) % 31,
                        torch.arange(n4) % 17], 1).to(torch.int32).to(dev)
    mask = (torch.rand(ncam, n4, generator=g) < 0.3).to(torch.uint8).to(dev)
    pos = torch.randint(0, max_ne + 40, (ncam, n4), generator=g).to(torch.int32).to(dev)      # some slots beyond max_ne
    f4 = torch.randn(n4, C, generator=g).to(dev)
    enh = torch.randn(B * ncam, max_ne, C, generator=g).to(dev)
    want = torch.empty_like(f4)
    _lib.check(lib.df3d_fusion_writeback(P(f4), P(enh), P(ind4), P(mask), P(pos), n4, C, ncam, max_ne, P(want), ops._stream()))
    got, gsplit = torch.empty_like(f4), torch.empty((n4, 4 * C), dtype=torch.uint8, device=dev)
    _lib.check(lib.df3d_fusion_writeback_split(P(f4), P(enh), P(ind4), P(mask), P(pos), n4, C, ncam, max_ne, P(got), P(gsplit),
                                               ops._stream()))
    assert torch.equal(got, want) and torch.equal(gsplit, ops.split_rows(want))
    # ---- image gate
    NI, Hf, Wf, Cs = B * ncam, 20, 33, 32
    gate = torch.randn(NI, Hf * Wf, generator=g).to(dev)
    b3 = torch.randn(1, generator=g).to(dev)
    S = torch.randn(NI, 9, Hf, Wf, generator=g).to(dev)
    kg = torch.randn(19, generator=g).to(dev)
    a0, a1 = torch.empty(NI, Hf, Wf, device=dev), torch.empty(NI, Hf, Wf, device=dev)
    _lib.check(lib.df3d_gate_finish(P((gate + b3).contiguous()), P(S), P(kg), NI, Hf, Wf, P(a0), ops._stream()))
    _lib.check(lib.df3d_gate_finish_bias(P(gate), P(b3), P(S), P(kg), NI, Hf, Wf, P(a1), ops._stream()))
    assert torch.equal(a0, a1)
    fs = torch.randn(n4, Cs, generator=g).to(dev)
    pinv = torch.randn(n4, 3, generator=g).to(dev)
    Tm = torch.randn(9, Cs + 3, generator=g).to(dev)
    grid = torch.stack([torch.randint(0, Wf, (ncam, n4), generator=g), torch.randint(0, Hf, (ncam, n4), generator=g)], 2) \
        .to(torch.int32).to(dev).contiguous()
    w0, S0 = torch.empty(NI, Hf, Wf, dtype=torch.int32, device=dev), torch.empty(NI, 9, Hf, Wf, device=dev)
    _lib.check(lib.df3d_gate_scatter_rows(P(fs), Cs, P(pinv), P(Tm), P(ind4), P(grid), P(mask), n4, B, ncam, Hf, Wf, P(w0), P(S0),
                                          1, ops._stream()))
    w1, S1 = torch.empty_like(w0), torch.empty_like(S0)
    _lib.check(lib.df3d_scatter_winner(P(ind4), P(grid), P(mask), n4, B, ncam, Hf, Wf, P(w1), ops._stream()))
    _lib.check(lib.df3d_gate_rows(P(fs), Cs, P(pinv), P(Tm), P(w1), NI, Hf, Wf, P(S1), 1, ops._stream()))
    assert torch.equal(w0, w1) and torch.equal(S0, S1) and int((w1 >= 0).sum()) > 100
