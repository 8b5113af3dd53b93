"""CPU tests of the host logic added in round 5: the Python side of the fp16 operand split (weight packing for the query
linears, the backward's precision context), the launch tape's purity check and the runtime-settings helper."""
import os
import sys
import warnings

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "3d-dual-fusion_amd")]


def test_weight_split_is_fp32_grade_and_range_checked():
    from dualfusion import _lib, ops
    g = torch.Generator().manual_seed(0)
    w = torch.randn(96, 128, generator=g) * torch.exp(torch.randn(96, 1, generator=g) * 2.0)
    w = w.clamp(-400.0, 400.0)
    w[0, :4] = torch.tensor([0.0, -0.0, 1e-9, 510.9])
    hi, lo = ops.split_weights_fp16(w)
    assert hi.dtype == lo.dtype == torch.float16
    rec = (hi.double() + lo.double()) / ops.SPLIT_W_SCALE
    err = (rec - w.double()).abs()
    assert bool((err <= torch.maximum(w.double().abs() * 2.0 ** -22, torch.tensor(2.0 ** -32, dtype=torch.float64))).all())
    # the same bits numpy's float16 conversion gives (round to nearest even, subnormals kept): what csrc/common.h computes
    ws = (w.numpy().astype(np.float32) * np.float32(ops.SPLIT_W_SCALE)).astype(np.float32)
    nh = ws.astype(np.float16)
    nl = (ws - nh.astype(np.float32)).astype(np.float16)
    assert np.array_equal(hi.numpy().view(np.uint16), nh.view(np.uint16))
    assert np.array_equal(lo.numpy().view(np.uint16), nl.view(np.uint16))
    for bad in (600.0, float("inf"), float("nan")):
        w2 = w.clone()
        w2[3, 7] = bad
        with pytest.raises(_lib.Df3dError):
            ops.split_weights_fp16(w2, "test")
    # the device constants mirror csrc/common.h
    src = open(os.path.join(ROOT, "3d-dual-fusion_amd", "csrc", "common.h")).read()
    assert "#define DF3D_SA_SCALE 32.0f" in src and "#define DF3D_SW_SCALE 128.0f" in src
    assert (ops.SPLIT_ACT_SCALE, ops.SPLIT_W_SCALE) == (32.0, 128.0)
    assert "#define DF3D_ACC_UNSCALE 0.000244140625f" in src and 1.0 / (32.0 * 128.0) == 0.000244140625


def test_unsplit_rows_inverts_the_activation_split():
    from dualfusion import ops
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(50, 64, generator=g) * 3.0).float()
    xs = x * ops.SPLIT_ACT_SCALE
    hi = xs.to(torch.float16)
    lo = (xs - hi.float()).to(torch.float16)
    rows = torch.stack([hi.view(50, 8, 8), lo.view(50, 8, 8)], 2).contiguous().view(torch.uint8).reshape(50, 256)
    back = ops.unsplit_rows(rows, 50, 64)
    assert float((back.double() - x.double()).abs().max()) <= 2.0 ** -22 * float(x.abs().max())


def test_backward_runs_gradient_convolutions_in_the_three_part_mode():
    from dualfusion import ops
    old = ops.CONV_PRECISION
    try:
        for mode, inside in (("split", "split3"), ("split3", "split3"), ("bf16", "bf16"), ("fp32", "fp32")):
            ops.CONV_PRECISION = mode
            with ops.grad_precision():
                assert ops.CONV_PRECISION == inside and ops.split_parts() == (3 if inside == "split3" else 2)
            assert ops.CONV_PRECISION == mode
        ops.CONV_PRECISION = "split"
        with pytest.raises(RuntimeError):
            with ops.grad_precision():
                raise RuntimeError("the mode is restored on the way out")
        assert ops.CONV_PRECISION == "split"
    finally:
        ops.CONV_PRECISION = old


def test_launch_tape_purity_check_tells_allocations_and_views_from_launches():
    from dualfusion.tape import _PurityMode
    m = _PurityMode()
    with m:
        a = torch.empty((4, 8))
        views = [a.view(8, 4), a.permute(1, 0), a.contiguous(), a[:, 0:4], a.new_empty((2,)), torch.empty_like(a),
                 a.view(torch.uint8), a.reshape(32), a.t(), a.unsqueeze(0), a.detach(), a.narrow(0, 0, 2), a.split(2, 0),
                 a.expand(4, 8), torch.as_strided(a, (2, 2), (1, 1))]
        assert a.data_ptr() and len(views) == 15
    assert m.impure == []
    m = _PurityMode()
    with m:
        a = torch.zeros((4, 8))
        b = torch.cat([a, a], 1)
        b.t().contiguous()
        a.fill_(1)
        a + 1
    found = " ".join(m.impure)
    for op in ("zeros", "cat", "clone", "fill_", "add"):
        assert op in found, (op, m.impure)


def test_configure_runtime_leaves_exported_settings_alone_and_warns_when_late(monkeypatch):
    import dualfusion
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "7")
    monkeypatch.setenv("HIP_FORCE_DEV_KERNARG", "0")
    with warnings.catch_warnings():
        warnings.simplefilter("error")                       # nothing is missing: no warning whatever torch's state is
        assert dualfusion.configure_runtime() is True
    assert os.environ["GPU_MAX_HW_QUEUES"] == "7" and os.environ["HIP_FORCE_DEV_KERNARG"] == "0"
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: True)
    with pytest.warns(RuntimeWarning):
        assert dualfusion.configure_runtime() is False
    assert os.environ["GPU_MAX_HW_QUEUES"] == "16"
    monkeypatch.delenv("GPU_MAX_HW_QUEUES")
    monkeypatch.setattr(torch.cuda, "is_initialized", lambda: False)
    assert dualfusion.configure_runtime() is True


def test_chunked_nms_host_loop_matches_the_dense_matrix_loop():
    """the iou3d_cuda NMS shims walk the suppression matrix in row blocks; same survivors in the same order as the
    reference's loop over the whole matrix (TF/mmdet3d/ops/iou3d/src/iou3d.cpp:127-143)."""
    from dualfusion.ext import iou3d_cuda
    rs = np.random.RandomState(0)
    n = 700
    m = rs.rand(n, n) < 0.02
    m = np.triu(m | m.T, 1)
    removed, want = np.zeros(n, bool), []
    for i in range(n):
        if not removed[i]:
            want.append(i)
            removed[i + 1:] |= m[i, i + 1:]
    keep = torch.zeros(n, dtype=torch.long)
    old = iou3d_cuda._NMS_ROWS
    try:
        for rows in (64, 2048):
            iou3d_cuda._NMS_ROWS = rows
            k = iou3d_cuda._greedy_keep(n, lambda i0, i1: torch.from_numpy(m[i0:i1, i0:]), keep)
            assert k == len(want) and keep[:k].tolist() == want
    finally:
        iou3d_cuda._NMS_ROWS = old


def test_linear_rows_module_is_a_drop_in_linear_on_the_host():
    """dualfusion.linear_rows.Linear (the weight gradient of the adapter's linear layers on a native kernel, GPU only): on the
    host it IS torch.nn.Linear -- same parameter names, same values, same gradients, state dicts interchangeable; the modules
    that used nn.Linear before round 5 still load the same checkpoints."""
    from dualfusion.linear_rows import Linear, linear
    from dualfusion.msda import MSDeformAttn
    torch.manual_seed(0)
    a, b = Linear(12, 7), torch.nn.Linear(12, 7)
    b.load_state_dict(a.state_dict())
    assert list(a.state_dict()) == list(b.state_dict()) == ["weight", "bias"]
    x = torch.randn(5, 3, 12, requires_grad=True)
    y = a(x)
    assert y.grad_fn is not None and "LinearFunction" not in type(y.grad_fn).__name__      # plain autograd on the host
    y.sum().backward()
    xb = x.detach().clone().requires_grad_(True)
    b(xb).sum().backward()
    assert torch.equal(y, b(xb)) and torch.equal(a.weight.grad, b.weight.grad) and torch.equal(x.grad, xb.grad)
    assert torch.equal(linear(x, a.weight, None), torch.nn.functional.linear(x, a.weight))
    m = MSDeformAttn(d_model=64, q_model=64, n_levels=1, n_heads=4, n_points=4)
    assert {"sampling_offsets.weight", "attention_weights.bias", "value_proj.weight", "output_proj.bias"} <= set(m.state_dict())
    assert all(isinstance(getattr(m, k), torch.nn.Linear) for k in ("sampling_offsets", "attention_weights", "value_proj", "output_proj"))


def test_stack_views_returns_a_view_of_equally_spaced_maps_and_a_copy_otherwise():
    """fusion._stack_views (training path of the adapter): per-camera maps that are slices of one tensor come back as one strided
    view (no 246 MB copy per step), anything else as torch.stack's copy; values identical either way."""
    from dualfusion.fusion import _stack_views
    base = torch.arange(6 * 4 * 3 * 5, dtype=torch.float32).view(6, 4, 3, 5)
    maps = [base[i] for i in range(6)]
    v = _stack_views(maps)
    assert v.data_ptr() == base.data_ptr() and torch.equal(v, base)
    assert torch.equal(_stack_views(maps[::2]), base[::2]) and _stack_views(maps[::2]).data_ptr() == base.data_ptr()
    other = [m.clone() for m in maps]
    c = _stack_views(other)
    assert torch.equal(c, base) and c.data_ptr() != other[0].data_ptr()
    mixed = [maps[0], maps[2], maps[3]]
    assert torch.equal(_stack_views(mixed), torch.stack(mixed, 0))
    assert torch.equal(_stack_views([base.permute(0, 1, 3, 2)[i] for i in range(6)]), base.permute(0, 1, 3, 2))
