"""Parity tests proper: the HIP path (through the C ABI, libdf3d_hip.so) against the oracle on
the same seeded inputs and against the golden fixtures generated from the reference.
Bit-exact for voxel indices / rulebooks (canonical order); fp32 results within
1e-3 (north_star), in practice ~1e-5.  Run with -m gpu on a MI355X."""
import os

import copy

import numpy as np
import pytest

import detgen
from make_golden import RB_BATCH, RB_CASES, RB_SHAPE
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a MI355X"
    from dualfusion import _lib
    lib = _lib.load()
    buf = __import__("ctypes").create_string_buffer(64)
    assert lib.df3d_device_arch(buf, 64) == 0
    assert buf.value.decode().startswith("gfx950"), buf.value
    return torch.device("cuda:0")


def T(a, dev, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t if dtype is None else t.to(dtype)


# ------------------------------------------------------------------------- voxelize
@pytest.mark.parametrize("tag", ["nocap", "cap", "mp3"])
def test_voxelize_golden(golden, dev, tag):
    from dualfusion import ops, synth
    g = golden("voxelize.npz")
    maxp, maxv = [int(x) for x in g["sw_%s_params" % tag]]
    v, c, n, mean = ops.hard_voxelize(T(g["sw_points"], dev), synth.NUSC_VOXEL, synth.NUSC_RANGE, maxp, maxv)
    assert np.array_equal(c.cpu().numpy(), g["sw_%s_coors" % tag])
    assert np.array_equal(n.cpu().numpy(), g["sw_%s_num" % tag])
    assert np.array_equal(v[:, 0].cpu().numpy(), g["sw_%s_first" % tag])
    ov, oc, on = orc.hard_voxelize(g["sw_points"], synth.NUSC_VOXEL, synth.NUSC_RANGE, maxp, maxv)
    assert np.array_equal(v.cpu().numpy(), ov)           # padded point tensor bit-exact
    np.testing.assert_allclose(mean.cpu().numpy(), orc.mean_vfe(ov, on), rtol=1e-6, atol=1e-6)


def test_voxelize_reference_test_vector(golden, dev):
    from dualfusion import ops
    g = golden("voxelize.npz")
    v, c, n, mean = ops.hard_voxelize(T(g["tg_points"], dev), [0.5, 0.5, 0.5], [0, -40, -3, 70.4, 40, 1], 1000, 20000)
    assert np.array_equal(c.cpu().numpy(), g["tg_expected_coors"])
    assert np.array_equal(n.cpu().numpy(), g["tg_expected_num"])


@pytest.mark.parametrize("break_at_cap", [True, False])
def test_voxelize_full_sweep_vs_oracle(dev, break_at_cap):
    from dualfusion import ops, synth
    pts = synth.nusc_sweep(seed=3)
    for maxv in (120000, 30000):
        v, c, n, mean = ops.hard_voxelize(T(pts, dev), synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, maxv,
                                          break_at_cap=break_at_cap)
        ov, oc, on = orc.hard_voxelize(pts, synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, maxv,
                                       "cpp" if break_at_cap else "numba")
        assert np.array_equal(c.cpu().numpy(), oc)
        assert np.array_equal(n.cpu().numpy(), on)
        assert np.array_equal(v.cpu().numpy(), ov)


def test_voxelize_edge_cases(dev):
    from dualfusion import ops, synth
    # no point in range
    pts = np.full((100, 5), 1e4, np.float32)
    v, c, n, mean = ops.hard_voxelize(T(pts, dev), synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, 1000)
    assert c.shape[0] == 0 and n.shape[0] == 0
    # every point in ONE voxel, more points than max_points
    pts = np.zeros((500, 5), np.float32)
    pts[:, 3] = np.arange(500)
    v, c, n, mean = ops.hard_voxelize(T(pts, dev), synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, 1000)
    assert c.shape[0] == 1 and int(n[0]) == 10
    assert np.array_equal(v[0, :, 3].cpu().numpy(), np.arange(10, dtype=np.float32))   # arrival order
    # points exactly on the upper range boundary are dropped, lower boundary kept
    pts = np.array([[54.0, 0, 0, 1, 0], [-54.0, 0, 0, 2, 0], [0, 0, 3.0, 3, 0], [0, 0, -5.0, 4, 0]], np.float32)
    v, c, n, mean = ops.hard_voxelize(T(pts, dev), synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, 1000)
    ov, oc, on = orc.hard_voxelize(pts, synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, 1000)
    assert np.array_equal(c.cpu().numpy(), oc) and np.array_equal(v.cpu().numpy(), ov)


# ------------------------------------------------------------------------- rulebook
def _hip_rulebook(ind_t, batch, shape, ks, st, pd, dl, subm):
    from dualfusion import ops
    if subm:
        grid = ops.grid_build(ind_t, batch, shape)
        nbr = ops.subm_neighbors(grid, ind_t, ks, dl)
        return ind_t, nbr, list(shape)
    oshape = orc.get_conv_output_size(shape, ks, st, pd, dl)
    grid = ops.grid_build(ind_t, batch, shape)
    outids, ogrid = ops.conv_out_indices(ind_t, batch, shape, oshape, ks, st, pd, dl)
    nbr = ops.conv_neighbors(grid, outids, ks, st, pd, dl)
    return outids, nbr, oshape


@pytest.mark.parametrize("name", sorted(RB_CASES))
def test_rulebook_and_conv_golden(golden, dev, name):
    from dualfusion import ops
    g = golden("rulebook_conv.npz")
    ks, st, pd, dl, subm, cin, cout = RB_CASES[name]
    ind = g["indices"]
    ind_t = T(ind, dev)
    outids, nbr, oshape = _hip_rulebook(ind_t, RB_BATCH, RB_SHAPE, ks, st, pd, dl, subm)
    pairs, num = ops.nbr_to_pairs(nbr, len(ind))
    # canonical-order comparison with the reference build's rulebook: bit-exact
    ref_out, ref_lists = orc.canonical_rulebook(g[name + "_outids"], g[name + "_pairs"], g[name + "_num"])
    my_out, my_lists = orc.canonical_rulebook(outids.cpu().numpy(), pairs.cpu().numpy(), num.cpu().numpy())
    assert np.array_equal(my_out, ref_out)
    assert np.array_equal(num.cpu().numpy(), g[name + "_num"])
    for a, b in zip(my_lists, ref_lists):
        assert np.array_equal(a, b)
    if not subm:  # strided outputs come out sorted by flat index (the reference GPU path's order)
        assert np.array_equal(outids.cpu().numpy(), ref_out)
    # convolution values: rows follow OUR output order -> map to the reference's order
    feats = detgen.randn("feat_" + name, (len(ind), cin))
    filt = detgen.randn("filt_" + name, (ks[0], ks[1], ks[2], cin, cout), 0.2)
    y = ops.sparse_conv_fused(T(feats, dev), T(filt, dev).reshape(-1, cin, cout), nbr, outids.shape[0]).cpu().numpy()
    ref_y = g[name + "_y"]
    if not subm:
        key = lambda o: [tuple(r) for r in o]
        pos = {k: i for i, k in enumerate(key(outids.cpu().numpy()))}
        perm = np.array([pos[k] for k in key(g[name + "_outids"])])
        y = y[perm]
    np.testing.assert_allclose(y, ref_y, rtol=1e-3, atol=1e-4)
    # the reference-format entry: rebuild nbr from pairs and convolve again
    nbr2 = ops.pairs_to_nbr(pairs.contiguous(), num, outids.shape[0])
    assert torch.equal(nbr2, nbr)


CONV_SHAPES = [(5, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (4, 16), (24, 40)]


@pytest.mark.parametrize("cin,cout", CONV_SHAPES)
@pytest.mark.parametrize("n_seeds", [4, 40])
def test_sparse_conv_vs_oracle(dev, cin, cout, n_seeds):
    """All tuned channel shapes + the generic fallback, small and multi-tile sizes, with the fused
    epilogue (bias, folded BN, residual, ReLU)."""
    from dualfusion import ops
    shape, batch = [9, 48, 48], 2
    ind = detgen.clustered_voxels("cv%d" % n_seeds, batch, shape, n_seeds=n_seeds, walk=200)
    ind_t = T(ind, dev)
    feats = detgen.randn("cvf%d_%d" % (cin, n_seeds), (len(ind), cin))
    filt = detgen.randn("cvw%d_%d" % (cin, cout), (3, 3, 3, cin, cout), 0.5 / np.sqrt(cin))
    bias = detgen.randn("cvb%d" % cout, (cout,), 0.1)
    scale = 1 + detgen.randn("cvs%d" % cout, (cout,), 0.1)
    shift = detgen.randn("cvh%d" % cout, (cout,), 0.1)
    for subm in (1, 0):
        ks, st, pd, dl = [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1]
        o_out, o_pairs, o_num, oshape = orc.get_indice_pairs(ind, batch, shape, ks, st, pd, dl, subm)
        outids, nbr, _ = _hip_rulebook(ind_t, batch, shape, ks, st, pd, dl, subm)
        n_out = outids.shape[0]
        assert n_out == len(o_out)
        res = detgen.randn("cvr%d" % cout, (n_out, cout))
        y = ops.sparse_conv_fused(T(feats, dev), T(filt, dev).reshape(-1, cin, cout), nbr, n_out, bias=T(bias, dev),
                                  scale=T(scale, dev), shift=T(shift, dev), residual=T(res, dev), relu=True)
        ref = orc.indice_conv(feats, filt, o_pairs, o_num, len(o_out), subm)
        if not subm:  # oracle rows are in first-touch order; ours sorted
            pos = {tuple(r): i for i, r in enumerate(o_out.tolist())}
            ref = ref[[pos[tuple(r)] for r in outids.cpu().numpy().tolist()]]
        ref = np.maximum((ref + bias) * scale + shift + res, 0)
        np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-3, atol=1e-4)


def _np_split(x, scale=32.0):
    """hi = fp16_rne(S x), lo = fp16_rne(S x - hi) as uint16 bit patterns (csrc/common.h split_pair_f16_ref; S = 2^5 for
    activations, 2^7 for filters)."""
    xs = (x.astype(np.float32) * np.float32(scale)).astype(np.float32)
    with np.errstate(over="ignore"):
        hi = xs.astype(np.float16)
        lo = (xs - hi.astype(np.float32)).astype(np.float16)
    return hi.view(np.uint16), lo.view(np.uint16)


def test_split_rows_bit_exact(dev):
    """The two-part operand format: fp16 hi + lo of 2^5 x, bit for bit against numpy's float16 conversion (round to nearest
    even, subnormals kept); 22 significand bits down to |x| = 2^-8, an absolute error <= 2^-30 below; a value outside
    fp16's range raises the sticky flag instead of passing silently."""
    from dualfusion import ops
    ops.split_overflow(reset=True)
    x = detgen.randn("splitrows", (1000, 64)) * np.exp(detgen.randn("splitrows_s", (1000, 1)) * 2).astype(np.float32)
    x = np.clip(x, -2000.0, 2000.0).astype(np.float32)
    x[0, :8] = [0.0, -0.0, 1.0, -1.0, 3.0e-39, 2046.9, -2047.0, 0.333333343]
    x[1, :4] = [1e-3, 2.0 ** -8, 3e-6, 1e-9]
    got = ops.split_rows(T(x, dev))
    assert ops.split_overflow(reset=True) == (False, "")
    raw = got.cpu().numpy().view(np.uint16).reshape(1000, 8, 2, 8)
    hi, lo = _np_split(x)
    assert np.array_equal(raw[:, :, 0], hi.reshape(1000, 8, 8))
    assert np.array_equal(raw[:, :, 1], lo.reshape(1000, 8, 8))
    rec = ops.unsplit_rows(got, 1000, 64).cpu().numpy()
    err = np.abs(rec.astype(np.float64) - x.astype(np.float64))
    assert np.all(err <= np.maximum(np.abs(x) * 2.0 ** -22, 2.0 ** -30))
    # out of range: reported, not silent (and reported once: the flag is sticky until it is reset)
    x[5, 3] = 3000.0
    ops.split_rows(T(x, dev))
    hit, where = ops.split_overflow(reset=True)
    assert hit and "spconv_split" in where
    with pytest.raises(Exception):
        ops.split_rows(T(x, dev))
        ops.check_split_overflow()
    x[5, 3] = float("nan")
    ops.split_rows(T(x, dev))
    assert ops.split_overflow(reset=True)[0]
    assert ops.split_overflow(reset=True) == (False, "")


@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (64, 64), (64, 128), (128, 128)])
@pytest.mark.parametrize("n_seeds", [4, 60])
def test_sparse_conv_split_precision(dev, cin, cout, n_seeds):
    """Split-precision kernel (fp16 hi/lo operands, 3 MFMA products, fp32 accumulate) against the float64
    contraction: <= 4e-6 of the output scale -- the grade of the exact-fp32 kernel (~1e-6; rounds 1-4 split into bf16
    parts: ~1e-5) --, fused epilogue included, and its emitted split rows equal split_rows(out) bit for bit."""
    from dualfusion import ops
    shape, batch = [9, 48, 48], 2
    ind = detgen.clustered_voxels("cs%d" % n_seeds, batch, shape, n_seeds=n_seeds, walk=200)
    ind_t = T(ind, dev)
    feats = detgen.randn("csf%d_%d" % (cin, n_seeds), (len(ind), cin))
    filt = detgen.randn("csw%d_%d" % (cin, cout), (27, cin, cout), 0.5 / np.sqrt(cin))
    bias = detgen.randn("csb%d" % cout, (cout,), 0.1)
    scale = 1 + detgen.randn("css%d" % cout, (cout,), 0.1)
    shift = detgen.randn("csh%d" % cout, (cout,), 0.1)
    packed = ops.conv_pack_weights(T(filt, dev))
    fsplit = ops.split_rows(T(feats, dev))
    for subm in (1, 0):
        ks, st, pd, dl = [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1]
        outids, nbr, _ = _hip_rulebook(ind_t, batch, shape, ks, st, pd, dl, subm)
        n_out = outids.shape[0]
        res = detgen.randn("csr%d" % cout, (n_out, cout))
        for tiles in (None, ops.conv_tiles(nbr, cin, cout)):
            y, ys = ops.sparse_conv_split(fsplit, packed, nbr, n_out, cin, cout, bias=T(bias, dev),
                                          scale=T(scale, dev), shift=T(shift, dev), residual=T(res, dev), relu=True,
                                          tiles=tiles)
            nb = nbr.cpu().numpy()
            acc = np.zeros((n_out, cout), np.float64)
            for k in range(27):
                m = nb[k] >= 0
                acc[m] += feats[nb[k][m]].astype(np.float64) @ filt[k].astype(np.float64)
            ref = np.maximum((acc + bias) * scale + shift + res, 0)
            err = np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max()
            assert err < 4e-6, err
            assert torch.equal(ys, ops.split_rows(y))
        y32 = ops.sparse_conv_fused(T(feats, dev), T(filt, dev), nbr, n_out, bias=T(bias, dev), scale=T(scale, dev),
                                    shift=T(shift, dev), residual=T(res, dev), relu=True)
        assert np.abs(y32.cpu().numpy() - ref).max() / np.abs(ref).max() < 5e-6


@pytest.mark.parametrize("cin,cout", [(128, 128), (64, 128), (128, 256)])
def test_loader_consumer_conv_kernel_forced_on_small_and_ragged_shapes(dev, cin, cout):
    """The loader / consumer LDS-DMA kernel normally serves layers of >= 190 workgroups; forced on (DF3D_OS_LC=1) it must
    give the register-gather kernel's result BIT FOR BIT (same summation order) on every shape: a single row, row counts
    around the 128-row tile, tiles whose last rows are padding, offsets without any pair in a tile, a 1-offset rulebook
    (fewer steps than ring stages), every epilogue combination, 256 columns as two blocks, a tiling order."""
    import os
    from dualfusion import ops
    shape, batch = [9, 48, 48], 2
    filt = detgen.randn("lcw%d_%d" % (cin, cout), (27, cin, cout), 0.5 / np.sqrt(cin))
    packed = ops.conv_pack_weights(T(filt, dev))
    packed1 = ops.conv_pack_weights(T(filt[13:14].copy(), dev))
    bias, scale, shift = (T(detgen.randn("lc%s%d" % (t, cout), (cout,), 0.2), dev) for t in "bsh")
    old = os.environ.get("DF3D_OS_LC")
    old_ks = os.environ.get("DF3D_OS_KSPLIT")
    os.environ["DF3D_OS_KSPLIT"] = "0"          # (the register-gather kernel's offset split sums in another order: own test)

    def both(fn):
        os.environ["DF3D_OS_LC"] = "0"
        a = fn()
        os.environ["DF3D_OS_LC"] = "1"
        b = fn()
        return a, b
    try:
        for n_seeds, walk in ((1, 1), (1, 40), (2, 64), (3, 90), (8, 200), (40, 200)):
            ind = detgen.clustered_voxels("lc%d_%d" % (n_seeds, walk), batch, shape, n_seeds=n_seeds, walk=walk)
            ind_t = T(ind, dev)
            feats = detgen.randn("lcf%d_%d_%d" % (cin, n_seeds, walk), (len(ind), cin))
            fsplit = ops.split_rows(T(feats, dev))
            for subm in (1, 0):
                outids, nbr, _ = _hip_rulebook(ind_t, batch, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], subm)
                n_out = outids.shape[0]
                res = T(detgen.randn("lcr%d_%d" % (cout, n_out), (n_out, cout)), dev)
                for kw in (dict(), dict(bias=bias, relu=True), dict(bias=bias, scale=scale, shift=shift, residual=res, relu=True)):
                    (y0, s0), (y1, s1) = both(lambda: ops.sparse_conv_split(fsplit, packed, nbr, n_out, cin, cout, **kw))
                    assert torch.equal(y0, y1) and torch.equal(s0, s1), (n_out, subm, sorted(kw))
                if subm:
                    order = torch.randperm(n_out, device=dev, dtype=torch.int64).to(torch.int32)
                    (y0, s0), (y1, s1) = both(lambda: ops.sparse_conv_split(fsplit, packed, nbr, n_out, cin, cout, bias=bias,
                                                                          order=order))
                    assert torch.equal(y0, y1) and torch.equal(s0, s1)
                    nb1 = nbr[13:14].contiguous()                                     # 1 x 1 x 1 "convolution": KB steps in all
                    (y0, _), (y1, _) = both(lambda: ops.sparse_conv_split(fsplit, packed1, nb1, n_out, cin, cout, bias=bias))
                    assert torch.equal(y0, y1)
                    want = feats.astype(np.float64) @ filt[13].astype(np.float64) + bias.cpu().numpy()
                    assert np.abs(y1.cpu().numpy() - want).max() / max(np.abs(want).max(), 1e-9) < 4e-6
    finally:
        if old_ks is None:
            os.environ.pop("DF3D_OS_KSPLIT", None)
        else:
            os.environ["DF3D_OS_KSPLIT"] = old_ks
        if old is None:
            os.environ.pop("DF3D_OS_LC", None)
        else:
            os.environ["DF3D_OS_LC"] = old


@pytest.mark.parametrize("cin,cout", [(256, 256), (128, 256)])
def test_offset_split_wave_groups_match_the_single_group_kernel(dev, cin, cout):
    """Round 4: the register-gather kernel with three wave groups per workgroup sharing a tile's offsets (small maps of the BEV
    neck, `spconv_os_split_kernel<..., KS = 3>`; partial sums meet in LDS) against the single-group launch: the same products
    in another summation order (<= 2e-6 of the output scale), split rows that are exactly the split of its own fp32 rows.
    Dense 3 x 3 maps around the 64-row tile (ragged last tile, border tiles without some offsets), a 2 x 2 transposed table
    (4 offsets: shares of 2 / 2 / 0), a 3-D rulebook with 27 offsets and sparse tiles, every epilogue combination."""
    import os
    from dualfusion import ops
    bias, scale, shift = (T(detgen.randn("ks%s%d" % (t, cout), (cout,), 0.2), dev) for t in "bsh")
    old = os.environ.get("DF3D_OS_KSPLIT")

    def both(fn):
        os.environ["DF3D_OS_KSPLIT"] = "0"
        a = fn()
        os.environ["DF3D_OS_KSPLIT"] = "1"
        b = fn()
        return a, b

    def check(fsplit, packed, nbr, n_out, tag):
        res = T(detgen.randn("ksr%d_%d" % (cout, n_out), (n_out, cout)), dev)
        for kw in (dict(), dict(bias=bias, relu=True), dict(bias=bias, scale=scale, shift=shift, residual=res, relu=True)):
            (y0, s0), (y1, s1) = both(lambda: ops.sparse_conv_split(fsplit, packed, nbr, n_out, cin, cout, **kw))
            sc = float(y0.abs().max())
            assert float((y0 - y1).abs().max()) <= 2e-6 * sc, (tag, n_out, sorted(kw), float((y0 - y1).abs().max()) / sc)
            assert torch.equal(s1, ops.split_rows(y1)), (tag, n_out, sorted(kw))
    try:
        for (B, H, W) in ((1, 90, 90), (1, 79, 83), (2, 75, 80), (1, 8, 8)):          # (the last one: below the split's range)
            nbr = ops.conv2d_neighbors(B, H, W, 3, 3, 1, 1, False, dev)[0]
            n = B * H * W
            filt = detgen.randn("ksw%d_%d" % (cin, cout), (9, cin, cout), 0.5 / np.sqrt(cin))
            feats = T(detgen.randn("ksf%d_%d" % (cin, n), (n, cin)), dev)
            check(ops.split_rows(feats), ops.conv_pack_weights(T(filt, dev)), nbr, n, "3x3")
        nbr = ops.conv2d_neighbors(1, 45, 45, 2, 2, 2, 0, True, dev)[0]             # 2 x 2 stride-2 transposed: 4 offsets
        filt = detgen.randn("ksw4_%d_%d" % (cin, cout), (4, cin, cout), 0.5 / np.sqrt(cin))
        feats = T(detgen.randn("ksf4_%d" % cin, (45 * 45, cin)), dev)
        check(ops.split_rows(feats), ops.conv_pack_weights(T(filt, dev)), nbr, nbr.shape[1], "2x2T")
        shape, batch = [9, 48, 48], 2
        ind = detgen.clustered_voxels("ks3d", batch, shape, n_seeds=40, walk=200)
        ind_t = T(ind, dev)
        _, nbr, _ = _hip_rulebook(ind_t, batch, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], 1)
        filt = detgen.randn("ksw27_%d_%d" % (cin, cout), (27, cin, cout), 0.5 / np.sqrt(cin))
        feats = T(detgen.randn("ksf27_%d" % cin, (len(ind), cin)), dev)
        check(ops.split_rows(feats), ops.conv_pack_weights(T(filt, dev)), nbr, len(ind), "3x3x3")
    finally:
        if old is None:
            os.environ.pop("DF3D_OS_KSPLIT", None)
        else:
            os.environ["DF3D_OS_KSPLIT"] = old


def test_rulebook_empty_and_single(dev):
    from dualfusion import ops
    shape, batch = [5, 8, 8], 1
    ind = np.array([[0, 2, 3, 4]], np.int32)
    outids, nbr, _ = _hip_rulebook(T(ind, dev), batch, shape, [3, 3, 3], [1, 1, 1], [1, 1, 1], [1, 1, 1], 1)
    nb = nbr.cpu().numpy()
    assert nb[13, 0] == 0 and (np.delete(nb[:, 0], 13) == -1).all()
    outids, nbr, _ = _hip_rulebook(T(ind, dev), batch, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], 0)
    o = orc.get_indice_pairs(ind, batch, shape, [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1], 0)
    assert np.array_equal(np.sort(outids.cpu().numpy(), 0), np.sort(o[0], 0))


def test_dense(dev):
    from dualfusion import ops
    shape, batch = [2, 30, 28], 3
    ind = detgen.clustered_voxels("dn", batch, shape, n_seeds=5, walk=150)
    f = detgen.randn("dnf", (len(ind), 128))
    d = ops.sparse_to_dense(T(f, dev), T(ind, dev), batch, shape)
    assert np.array_equal(d.cpu().numpy(), orc.dense(f, ind, shape, batch))


# ------------------------------------------------------------------------- MSDA
def _msda(dev, value, shapes, loc, aw):
    from dualfusion import ops
    shp = torch.as_tensor(np.asarray(shapes), dtype=torch.long, device=dev)
    lsi = torch.cat((shp.new_zeros((1,)), shp.prod(1).cumsum(0)[:-1]))
    return ops.ms_deform_attn_forward(T(value, dev), shp, lsi, T(loc, dev), T(aw, dev)).cpu().numpy()


def _softmax(x):
    return torch.softmax(torch.from_numpy(x), -1).numpy()


def test_msda_golden(golden, dev):
    g = golden("msda.npz")
    # the reference's own check tolerances for fp32 (ops/test.py:57): rtol 1e-2, atol 1e-3; we hold 1e-5
    y = _msda(dev, g["t_value"], g["t_shapes"], g["t_loc"], g["t_aw"])
    np.testing.assert_allclose(y, g["t_out"], rtol=1e-4, atol=1e-7)
    N, M, D, Lq, L, P, H, W = [int(x) for x in g["h_dims"]]
    value = detgen.randn("msda_h_value", (N, H * W, M, D))
    loc = detgen.rand("msda_h_loc", (N, Lq, M, L, P, 2), -0.15, 1.15)
    aw = _softmax(detgen.randn("msda_h_aw", (N, Lq, M, L * P))).reshape(N, Lq, M, L, P)
    np.testing.assert_allclose(_msda(dev, value, [(H, W)], loc, aw), g["h_out"], rtol=1e-3, atol=2e-5)
    N, M, D, Lq, L, P = [int(x) for x in g["m_dims"]]
    shp = [tuple(int(v) for v in r) for r in g["m_shapes"]]
    S = sum(h * w for h, w in shp)
    value = detgen.randn("msda_m_value", (N, S, M, D))
    loc = detgen.rand("msda_m_loc", (N, Lq, M, L, P, 2), -0.1, 1.1)
    aw = _softmax(detgen.randn("msda_m_aw", (N, Lq, M, L * P))).reshape(N, Lq, M, L, P)
    np.testing.assert_allclose(_msda(dev, value, shp, loc, aw), g["m_out"], rtol=1e-3, atol=2e-5)


@pytest.mark.parametrize("D", [4, 8, 16, 32, 5])
def test_msda_vs_oracle_shapes(dev, D):
    N, M, Lq, L, P = 2, 8, 333, 2, 3
    shp = [(17, 23), (9, 11)]
    S = sum(h * w for h, w in shp)
    value = detgen.randn("mo_v%d" % D, (N, S, M, D))
    loc = detgen.rand("mo_l%d" % D, (N, Lq, M, L, P, 2), -0.2, 1.2)
    aw = _softmax(detgen.randn("mo_a%d" % D, (N, Lq, M, L * P))).reshape(N, Lq, M, L, P)
    np.testing.assert_allclose(_msda(dev, value, shp, loc, aw), orc.ms_deform_attn(value, shp, loc, aw),
                               rtol=1e-3, atol=2e-5)


@pytest.mark.parametrize("D,L,P,wide", [(16, 1, 4, False), (16, 1, 4, True), (8, 2, 3, False), (32, 3, 5, False),
                                        (4, 1, 1, False)])
def test_msda_fused_vs_oracle(dev, D, L, P, wide):
    """Raw offsets/logits + 2-D reference points in, softmax and location arithmetic in the kernel
    (ms_deform_attn.py:149-166); `wide` samples a channel slice of a 2x wider value buffer."""
    from dualfusion import ops
    N, M, Lq = 2, 8, 257
    shp = [(17, 23), (9, 11), (5, 4)][:L]
    S = sum(h * w for h, w in shp)
    tag = "mf%d_%d_%d" % (D, L, P)
    value = detgen.randn(tag + "v", (N, S, M, D))
    ref = detgen.rand(tag + "r", (N, Lq, 2), -0.05, 1.05)
    off = detgen.randn(tag + "o", (N, Lq, M, L, P, 2)) * 2.0
    lg = detgen.randn(tag + "l", (N, Lq, M, L * P)) * 3.0
    norm = np.array([[w, h] for h, w in shp], np.float32)
    loc = (ref[:, :, None, None, None, :] + off / norm[None, None, None, :, None, :]).astype(np.float32)
    aw = _softmax(lg).reshape(N, Lq, M, L, P)
    want = orc.ms_deform_attn(value, shp, loc, aw)
    shapes = torch.as_tensor(shp, dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    v = T(value, dev)
    if wide:
        buf = torch.full((N, S, 2, M, D), 7.0, device=dev)
        buf[:, :, 1] = v
        v = buf[:, :, 1]
    y = ops.ms_deform_attn_fused(v, shapes, lsi, T(ref, dev), T(off.reshape(N, Lq, -1), dev), T(lg, dev), L, P)
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=1e-3, atol=2e-5)


@pytest.mark.parametrize("D,L,P,wide", [(16, 1, 4, False), (16, 1, 4, True), (8, 2, 3, False)])
def test_msda_fused_bf16_value(dev, D, L, P, wide):
    """The sampler on bf16 value rows (reduced-precision mode): against the oracle evaluated on the SAME bf16-rounded values
    it agrees to fp32 noise (weights, accumulation and output stay fp32); with a per-pixel scale and an image constant too."""
    from dualfusion import ops
    N, M, Lq = 2, 8, 257
    shp = [(17, 23), (9, 11)][:L]
    S = sum(h * w for h, w in shp)
    tag = "mfb%d_%d_%d" % (D, L, P)
    value = detgen.randn(tag + "v", (N, S, M, D))
    ref = detgen.rand(tag + "r", (N, Lq, 2), -0.05, 1.05)
    off = detgen.randn(tag + "o", (N, Lq, M, L, P, 2)) * 2.0
    lg = detgen.randn(tag + "l", (N, Lq, M, L * P)) * 3.0
    norm = np.array([[w, h] for h, w in shp], np.float32)
    loc = (ref[:, :, None, None, None, :] + off / norm[None, None, None, :, None, :]).astype(np.float32)
    aw = _softmax(lg).reshape(N, Lq, M, L, P)
    v16 = T(value, dev).to(torch.bfloat16)
    rounded = v16.float().cpu().numpy()
    shapes = torch.as_tensor(shp, dtype=torch.long, device=dev)
    lsi = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    v = v16
    if wide:
        buf = torch.full((N, S, 2, M, D), 7.0, device=dev, dtype=torch.bfloat16)
        buf[:, :, 1] = v16
        v = buf[:, :, 1]
    y = ops.ms_deform_attn_fused(v, shapes, lsi, T(ref, dev), T(off.reshape(N, Lq, -1), dev), T(lg, dev), L, P)
    np.testing.assert_allclose(y.cpu().numpy(), orc.ms_deform_attn(rounded, shp, loc, aw), rtol=1e-3, atol=2e-5)
    scale = detgen.rand(tag + "s", (N, S), 0.2, 1.5)
    cb = detgen.randn(tag + "c", (N, M * D))
    y = ops.ms_deform_attn_fused(v, shapes, lsi, T(ref, dev), T(off.reshape(N, Lq, -1), dev), T(lg, dev), L, P,
                                 pixel_scale=T(scale, dev), image_bias=T(cb, dev))
    want = orc.ms_deform_attn(rounded * scale[:, :, None, None] + cb.reshape(N, 1, M, D), shp, loc, aw)
    np.testing.assert_allclose(y.cpu().numpy(), want, rtol=1e-3, atol=5e-5)


@pytest.mark.parametrize("C,rows", [(128, 1000), (64, 77), (256, 3), (1024, 5), (16, 130), (20, 9), (100, 31), (132, 6)])
def test_actr_rowwise_kernels(dev, C, rows):
    """actr_prep / add_layernorm / bigate_sum against the torch expressions the reference layer runs
    (actr_transformer.py:399-426, attentions.py:96-117), fp32 within 1e-5."""
    from dualfusion import ops
    g = torch.Generator(device="cpu").manual_seed(C * 1000 + rows)
    r = lambda *s: torch.randn(*s, generator=g).to(dev)
    q, qi, pos = r(2, rows, C), r(2, rows, C), r(2, rows, C)
    A, Bw = ops.actr_prep(q, qi, pos)
    assert torch.equal(A, q + pos) and torch.equal(Bw, (q + pos) + (qi + pos))
    w, b = r(C), r(C)
    want = torch.nn.functional.layer_norm(q + qi, (C,), w, b, 1e-5)
    torch.testing.assert_close(ops.add_layernorm(q, qi, w, b, 1e-5), want, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ops.add_layernorm(q, None, w, b, 1e-5),
                               torch.nn.functional.layer_norm(q, (C,), w, b, 1e-5), rtol=1e-5, atol=1e-5)
    wb, wa, bb, ba = r(C) * 0.2, r(C) * 0.2, r(1), r(1)
    fuse = q + qi
    s1 = torch.sigmoid(fuse @ wb + bb)[..., None]
    s2 = torch.sigmoid(fuse @ wa + ba)[..., None]
    qo, qio = ops.bigate_sum(q, qi, wb, bb, wa, ba)
    torch.testing.assert_close(qo, q + qi * s1, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(qio, qi + q * s2, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("gated", [True, False])
def test_groupnorm_fold(dev, gated):
    """value_proj(GroupNorm(gate * u + b)) == gate * (Wf u) + cf  (actr.py:139-149 + ms_deform_attn.py:139),
    and the sampler's pixel_scale / image_bias inputs reproduce sampling the materialised map."""
    from dualfusion import ops
    g = torch.Generator(device="cpu").manual_seed(5 + gated)
    N, C, H, W, O = 3, 128, 13, 17, 256
    S = H * W
    u = (torch.randn(N, C + 1, S, generator=g) * 1.7 + 0.3).to(dev)
    gate = torch.rand(N, S, generator=g).to(dev) if gated else None
    gn = torch.nn.GroupNorm(32, C).to(dev)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C, generator=g))
        gn.bias.copy_(torch.randn(C, generator=g))
    b = torch.randn(C, generator=g).to(dev)
    Wv = (torch.randn(O, C, generator=g) * 0.1).to(dev)
    wb = torch.randn(O, generator=g).to(dev)
    with torch.no_grad():
        Wf, cf = ops.groupnorm_fold(u, gate, b, gn, Wv, wb)
        x = u[:, :C] * (gate[:, None] if gated else 1.0) + b[None, :, None]
        want = torch.einsum('oc,ncs->nso', Wv.double(), gn(x).double()) + wb.double()
        raw = torch.bmm(u[:, :C].transpose(1, 2), Wf.transpose(1, 2))
        got = raw * (gate[..., None] if gated else 1.0) + cf[:, None]
    torch.testing.assert_close(got, want.float(), rtol=1e-4, atol=1e-4)
    # sampler with the scale / constant applied on the fly == sampler on the materialised value (layer 1 slice)
    M, D, Lq, L, P = 8, 16, 301, 1, 4
    shapes = torch.as_tensor([(H, W)], dtype=torch.long, device=dev)
    lsi = shapes.new_zeros((1,))
    ref = (torch.rand(N, Lq, 2, generator=g) * 1.1 - 0.05).to(dev)
    off = (torch.randn(N, Lq, M * L * P * 2, generator=g) * 2).to(dev)
    lg = torch.randn(N, Lq, M * L * P, generator=g).to(dev)
    v_mat = got[:, :, 128:].contiguous().view(N, S, M, D)
    y0 = ops.ms_deform_attn_fused(v_mat, shapes, lsi, ref, off, lg, L, P)
    y1 = ops.ms_deform_attn_fused(raw[:, :, 128:].unflatten(-1, (M, D)), shapes, lsi, ref, off, lg, L, P,
                                  gate, cf[:, 128:])
    torch.testing.assert_close(y1, y0, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("rows,H,ln,res,C", [(1000, 1024, True, True, 128), (37, 256, True, False, 128),
                                              (130, 128, False, True, 128), (31000, 1024, True, True, 128),
                                              (45001, 256, True, True, 128),      # two row tiles per wave (>= 160 workgroups)
                                              (1000, 1024, True, True, 64), (37, 128, False, True, 64),
                                              (40001, 128, False, True, 64), (513, 256, True, False, 64)])
def test_ffn_fused(dev, rows, H, ln, res, C):
    """LayerNorm(x + W2 relu(W1 x + b1) + b2) in one split-precision kernel against the float64 composition
    (actr_transformer.py:413-424); 5e-5 of the output scale (parity bar 1e-3)."""
    from dualfusion import ops
    g = torch.Generator(device="cpu").manual_seed(rows + H + C)
    x = (torch.randn(rows, C, generator=g) * 1.3).to(dev)
    w1 = (torch.randn(H, C, generator=g) / C ** 0.5).to(dev)
    w2 = (torch.randn(C, H, generator=g) / H ** 0.5).to(dev)
    b1, b2 = torch.randn(H, generator=g).to(dev) * 0.3, torch.randn(C, generator=g).to(dev) * 0.3
    lw, lb = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    packed = ops.ffn_pack(w1, w2)
    y = ops.ffn_fused(x, packed, b1, b2, H, residual=x if res else None, ln_weight=lw if ln else None,
                      ln_bias=lb if ln else None, eps=1e-5)
    xd = x.double()
    ref = torch.relu(xd @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
    if res:
        ref = ref + xd
    if ln:
        ref = torch.nn.functional.layer_norm(ref, (C,), lw.double(), lb.double(), 1e-5)
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    assert err < 4e-6, err                              # fp32-grade (fp16 hi + lo operands; rounds 1-4, bf16 parts: 5e-5)


@pytest.mark.parametrize("rows", [37, 5000, 40001])
def test_ffn_fused_bf16_mode(dev, rows):
    """The bf16 mode of the fused feed-forward kernel (BASELINE configs[2]: bf16 operands, fp32 accumulate): against the
    float64 composition evaluated on the SAME bf16-rounded operands (x, W1, the hidden activation, W2) the result agrees
    to fp32 summation noise; against the unrounded fp32 layer it sits at bf16 level (a few 1e-3 of the output scale)."""
    from dualfusion import ops
    g = torch.Generator(device="cpu").manual_seed(rows)
    C, H = 128, 1024
    x = (torch.randn(rows, C, generator=g) * 1.3).to(dev)
    w1 = (torch.randn(H, C, generator=g) / C ** 0.5).to(dev)
    w2 = (torch.randn(C, H, generator=g) / H ** 0.5).to(dev)
    b1, b2 = torch.randn(H, generator=g).to(dev) * 0.3, torch.randn(C, generator=g).to(dev) * 0.3
    lw, lb = torch.randn(C, generator=g).to(dev), torch.randn(C, generator=g).to(dev)
    packed = ops.ffn_pack(w1, w2)
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = "bf16"
    try:
        y = ops.ffn_fused(x, packed, b1, b2, H, residual=x, ln_weight=lw, ln_bias=lb, eps=1e-5)
    finally:
        ops.CONV_PRECISION = old
    y32 = ops.ffn_fused(x, packed, b1, b2, H, residual=x, ln_weight=lw, ln_bias=lb, eps=1e-5)      # split precision again
    r = lambda t: t.to(torch.float16).double()         # noqa: E731  (round to nearest even, like the kernel's hi part)
    hid = torch.relu(r(x) @ r(w1).t() + b1.double())
    ref = r(hid.float()) @ r(w2).t() + b2.double() + x.double()
    ref = torch.nn.functional.layer_norm(ref, (C,), lw.double(), lb.double(), 1e-5)
    err = float((y.double() - ref).abs().max() / ref.abs().max())
    assert err < 5e-5, err                              # hidden values that round across an fp16 boundary differ by 1 ulp
    full = torch.relu(x.double() @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double() + x.double()
    full = torch.nn.functional.layer_norm(full, (C,), lw.double(), lb.double(), 1e-5)
    assert float((y.double() - full).abs().max() / full.abs().max()) < 3e-3
    assert float((y32.double() - full).abs().max() / full.abs().max()) < 4e-6          # the switch went back


@pytest.mark.parametrize("N,Q,C,G", [(6, 5189, 128, 32), (2, 77, 256, 32), (1, 1, 64, 16), (3, 300, 128, 8)])
def test_rows_groupnorm(dev, N, Q, C, G):
    """GroupNorm on the [N, Q, C] layout == the reference's GroupNorm on the Conv1d layout [N, C, Q]."""
    from dualfusion import ops
    g = torch.Generator(device="cpu").manual_seed(N * 1000 + Q)
    x = (torch.randn(N, Q, C, generator=g) * 2.0 + 0.7).to(dev)
    gn = torch.nn.GroupNorm(G, C).to(dev)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C, generator=g))
        gn.bias.copy_(torch.randn(C, generator=g))
        want = gn(x.transpose(1, 2)).transpose(1, 2)
        got = ops.rows_groupnorm(x, gn)
    torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("S,gated", [(40050, True), (1000, False), (77, True)])
def test_image_projection_split_path(dev, S, gated):
    """csrc/imgproj.hip: u = Wcat.img from channel-first maps (split rows + gate row) and the folded value GEMM
    against float64: W GroupNorm(att*u + b) + wb == att * value + cf, 5e-5 of the output scale."""
    from dualfusion import ops
    g = torch.Generator(device="cpu").manual_seed(S)
    NI, Cin, C = 3, 256, 128
    maps = [(torch.randn(Cin, S, generator=g) * 1.5).to(dev) for _ in range(NI)]
    wcat = (torch.randn(129, Cin, generator=g) / Cin ** 0.5).to(dev)
    ptrs = torch.tensor([m.data_ptr() for m in maps], dtype=torch.int64, device=dev)
    us, gate = ops.imgproj_split(ptrs, NI, Cin, S, ops.imgproj_pack(wcat))
    ref_u = torch.stack([wcat.double() @ m.double() for m in maps])             # [NI, 129, S]
    # split rows -> floats
    u_got = ops.unsplit_rows(us, NI * S, C).reshape(NI, S, C)
    scale = ref_u.abs().max()
    assert float((u_got.double() - ref_u[:, :C].transpose(1, 2)).abs().max() / scale) < 4e-6
    assert float((gate.double() - ref_u[:, C]).abs().max() / scale) < 4e-6
    att = torch.rand(NI, S, generator=g).to(dev) if gated else None
    gn = torch.nn.GroupNorm(32, C).to(dev)
    with torch.no_grad():
        gn.weight.copy_(torch.randn(C, generator=g))
        gn.bias.copy_(torch.randn(C, generator=g))
    b = torch.randn(C, generator=g).to(dev)
    Wv = (torch.randn(256, C, generator=g) * 0.1).to(dev)
    wb = torch.randn(256, generator=g).to(dev)
    with torch.no_grad():
        x = ref_u[:, :C].float() * (att[:, None] if gated else 1.0) + b[None, :, None]
        want = torch.einsum('oc,ncs->nso', Wv.double(), gn(x).double()) + wb.double()
        value, cf = ops.value_fold_gemm(us, att, b, gn, Wv, wb)
        got = value.double() * (att[..., None].double() if gated else 1.0) + cf[:, None].double()
    assert float((got - want).abs().max() / want.abs().max()) < 4e-6


def test_msda_linearity_at_full_size(dev):
    """BASELINE config 2 size (6 cams, 150x267 map, Q=8000): linear in value and in the weights."""
    from dualfusion import ops
    N, M, D, Lq, L, P, H, W = 6, 8, 16, 8000, 1, 4, 150, 267
    g = torch.Generator(device="cpu").manual_seed(0)
    v1 = torch.randn(N, H * W, M, D, generator=g).to(dev)
    v2 = torch.randn(N, H * W, M, D, generator=g).to(dev)
    loc = (torch.rand(N, Lq, M, L, P, 2, generator=g) * 1.2 - 0.1).to(dev)
    aw = torch.softmax(torch.randn(N, Lq, M, L * P, generator=g), -1).view(N, Lq, M, L, P).to(dev)
    shp = torch.as_tensor([(H, W)], dtype=torch.long, device=dev)
    lsi = shp.new_zeros((1,))
    f = lambda v, a: ops.ms_deform_attn_forward(v.contiguous(), shp, lsi, loc, a.contiguous())
    torch.testing.assert_close(f(v1 + 2 * v2, aw), f(v1, aw) + 2 * f(v2, aw), rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(f(v1, 0.5 * aw), 0.5 * f(v1, aw), rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------- point ops
def test_fps_ball_group_vs_oracle(dev):
    from dualfusion import ops
    B, N, m, ns = 3, 3000, 256, 16
    xyz = detgen.rand("fps_xyz", (B, N, 3), -20, 20)
    xyz[1, 2500:] = 0  # zero-padded tail rows (duplicates -> FPS ties), as the ACTR adapter pads
    idx = ops.furthest_point_sample(T(xyz, dev), m).cpu().numpy()
    assert np.array_equal(idx, orc.furthest_point_sample(xyz, m))
    new_xyz = np.stack([xyz[b][idx[b]] for b in range(B)])
    bq = ops.ball_query(0.0, 4.0, ns, T(xyz, dev), T(new_xyz, dev)).cpu().numpy()
    assert np.array_equal(bq, orc.ball_query(0.0, 4.0, ns, xyz, new_xyz))
    feat = detgen.randn("fps_feat", (B, 24, N))
    gp = ops.group_points(T(feat, dev), T(bq, dev)).cpu().numpy()
    assert np.array_equal(gp, orc.group_points(feat, bq))
    ga = ops.gather_points(T(feat, dev), T(idx, dev)).cpu().numpy()
    assert np.array_equal(ga, orc.gather_points(feat, idx))


@pytest.mark.parametrize("N", [50, 64, 1000, 5000, 12000, 17000, 20001, 24576, 26000])
def test_fps_block_sizes(dev, N):
    """every kernel variant (points per thread 8/16/24-as-half-block/32, global fallback) against the oracle,
    with a zero-padded tail (exact distance ties, resolved by the reference's thread-id rule)"""
    from dualfusion import ops
    xyz = detgen.rand("fpsb%d" % N, (2, N, 3), -5, 5)
    if N >= 1000:
        xyz[1, N - N // 5:] = 0
        xyz[0, ::7] = xyz[0, 3]          # duplicated points spread over many threads
    m = min(N, 40 if N < 17000 else 300)      # the half-block kernel (16k < N <= 24k) runs long enough to alternate its buffers
    assert np.array_equal(ops.furthest_point_sample(T(xyz, dev), m).cpu().numpy(), orc.furthest_point_sample(xyz, m))


# ------------------------------------------------------------------------- MSDA backward
@pytest.mark.parametrize("tag", ["hot", "multi"])
def test_msda_backward_golden_and_oracle(golden, dev, tag):
    """df3d_ms_deform_attn_backward against the reference's autograd gradients (float64 pure-torch core) and the oracle;
    through MSDeformAttnFunction.backward, i.e. the way the reference's modules reach it."""
    from dualfusion.msda import MSDeformAttnFunction
    from make_golden import msda_bwd_inputs
    g = golden("msda_bwd.npz")
    value, shp, loc, aw, gout = msda_bwd_inputs(tag)
    v, lo, a = (T(x, dev).requires_grad_(True) for x in (value, loc, aw))
    shapes = torch.as_tensor(shp, dtype=torch.long, device=dev)
    lstart = torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))
    y = MSDeformAttnFunction.apply(v, shapes, lstart, lo, a, 64)
    y.backward(T(gout, dev))
    for got, key in ((v.grad, "_gv"), (lo.grad, "_gl"), (a.grad, "_ga")):
        want = g[tag + key]
        err = np.abs(got.cpu().numpy() - want).max() / max(1.0, np.abs(want).max())
        assert err <= 1e-4, (key, err)                         # fp32 kernel vs float64 reference; measured ~1e-6


def test_msda_backward_full_size_properties(dev):
    """6 cameras x 150x267 map, 8000 queries (the CenterPoint fusion layer): linearity in grad_output, and the
    directional derivative <grad_value, dV> = d/dt <out(V + t dV), g> (out is linear in value: exact up to rounding)."""
    from dualfusion import ops
    N, H, W, M, D, Lq, P = 6, 150, 267, 8, 16, 8000, 4
    gen = torch.Generator(device="cpu").manual_seed(5)
    value = torch.randn(N, H * W, M, D, generator=gen).to(dev)
    loc = (torch.rand(N, Lq, M, 1, P, 2, generator=gen) * 1.2 - 0.1).to(dev)
    aw = torch.softmax(torch.randn(N, Lq, M, P, generator=gen), -1).view(N, Lq, M, 1, P).to(dev)
    g1 = torch.randn(N, Lq, M * D, generator=gen).to(dev)
    g2 = torch.randn(N, Lq, M * D, generator=gen).to(dev)
    shapes = torch.as_tensor([(H, W)], dtype=torch.long, device=dev)
    lstart = shapes.new_zeros((1,))
    a = ops.ms_deform_attn_backward(value, shapes, lstart, loc, aw, g1)
    b = ops.ms_deform_attn_backward(value, shapes, lstart, loc, aw, g2)
    c = ops.ms_deform_attn_backward(value, shapes, lstart, loc, aw, g1 + 2 * g2)
    for x, y, z in zip(a, b, c):
        assert float((z - (x + 2 * y)).abs().max()) <= 2e-4 * float(z.abs().max())
    dv = torch.randn(N, H * W, M, D, generator=gen).to(dev)
    lhs = float((a[0].double() * dv.double()).sum())
    rhs = float((ops.ms_deform_attn_forward(dv, shapes, lstart, loc, aw).double() * g1.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(rhs))
    # grad_attn_weight is the forward of each sample alone: <out, g> = sum_p aw_p * grad_aw_p
    out = ops.ms_deform_attn_forward(value, shapes, lstart, loc, aw)
    lhs = float((out.double() * g1.double()).sum())
    rhs = float((a[2].double() * aw.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(rhs))


# ------------------------------------------------------------------------- bf16 convolution (BASELINE configs[2])
@pytest.mark.parametrize("cin,cout,n_seeds,ks,st,subm", [(128, 128, 14, 3, 1, 1), (64, 64, 9, 3, 1, 1), (32, 64, 9, 3, 2, 0),
                                                        (64, 128, 6, 3, 2, 0), (32, 32, 4, 3, 1, 1), (128, 64, 6, 3, 1, 1),
                                                        (64, 32, 9, 3, 1, 1)])
def test_conv_bf16_equals_fp32_kernel_on_rounded_operands(dev, cin, cout, n_seeds, ks, st, subm):
    """df3d_sparse_conv_bf16 (bf16 rows, bf16 weights, fp32 accumulate, fused BN / residual / ReLU) against the exact
    fp32 kernel fed the SAME bf16-rounded operands: products of bf16 numbers are exact in fp32, so only the
    accumulation order differs (<= 2e-5 of the output scale); the bf16 output rows are the RNE rounding of it."""
    from dualfusion import ops
    from dualfusion.spconv import ops as sops
    shape, batch = [9, 40, 44], 2
    ind = detgen.clustered_voxels("bfc%d" % n_seeds, batch, shape, n_seeds=n_seeds, walk=220)
    it = T(ind, dev)
    outids, nbr, _, _ = sops.build_rulebook(it, batch, shape, [ks] * 3, [st] * 3, [1] * 3, [1] * 3, bool(subm))
    n_out = outids.shape[0]
    f = T(detgen.randn("bfc_f%d_%d" % (cin, n_seeds), (len(ind), cin)), dev)
    w = T(detgen.randn("bfc_w%d_%d" % (cin, cout), (ks ** 3, cin, cout), 0.1), dev)
    bias, scale, shift = (T(detgen.randn("bfc_%s%d" % (nm, cout), (cout,), 0.3), dev) for nm in "bsh")
    scale = scale + 1.0
    res = T(detgen.randn("bfc_r%d_%d" % (cout, n_seeds), (n_out, cout)), dev)
    fb, rb = ops.rows_to_bf16(f), ops.rows_to_bf16(res)
    assert torch.equal(fb, f.to(torch.bfloat16)) and torch.equal(ops.rows_from_bf16(fb), fb.float())
    wb = w.to(torch.bfloat16).float()
    want = ops.sparse_conv_fused(fb.float(), wb, nbr, n_out, bias=bias, scale=scale, shift=shift, residual=rb.float(),
                                 relu=True)
    got32, got16 = ops.sparse_conv_bf16(fb, ops.conv_pack_weights_bf16(w), nbr, n_out, cin, cout, bias=bias, scale=scale,
                                        shift=shift, residual=rb, relu=True, want_f32=True, want_bf16=True)
    err = float((got32 - want).abs().max() / want.abs().max())
    assert err <= 2e-5, err
    assert torch.equal(got16, got32.to(torch.bfloat16))
    # bf16-only output (no fp32 copy) and no epilogue operands
    _, only16 = ops.sparse_conv_bf16(fb, ops.conv_pack_weights_bf16(w), nbr, n_out, cin, cout)
    plain = ops.sparse_conv_fused(fb.float(), wb, nbr, n_out)
    assert float((only16.float() - plain).abs().max() / plain.abs().max()) <= 5e-3      # bf16 rounding of the result
    # vs the fp32 operands: the bf16 path's error is the operand rounding (~2^-9 per term, averaging out)
    full = ops.sparse_conv_fused(f, w, nbr, n_out)
    assert float((plain - full).abs().max() / full.abs().max()) <= 2e-2


# ------------------------------------------------------------------------- sparse max pool, inverse conv, dynamic voxelize
def _canon(ids):
    return np.lexsort(np.asarray(ids).T[::-1])


def test_sparse_maxpool_module_golden_and_oracle(golden, dev):
    """SparseMaxPool3d forward / backward (one launch each over the neighbour table) against the reference's compiled
    CPU outputs (tests/golden/pool.npz) and the oracle -- bit for bit, incl. the zero-initialised output and the
    fan-out of the gradient over equal values."""
    from dualfusion import spconv
    from make_golden import CONV_BWD_BATCH, CONV_BWD_SHAPE, conv_bwd_case
    g = golden("pool.npz")
    ind, ks, st, pd, f, _ = conv_bwd_case(0)
    pool = spconv.SparseMaxPool3d(ks, st, pd)
    for feats, ykey, gkey in ((f, "y", "gin"), (np.round(f * 2) / 2, "yq", "ginq")):
        x = T(feats, dev).requires_grad_(True)
        out = pool(spconv.SparseConvTensor(x, T(ind, dev), CONV_BWD_SHAPE, CONV_BWD_BATCH))
        o = _canon(out.indices.cpu().numpy())
        assert np.array_equal(out.indices.cpu().numpy()[o], g["outids"]) and out.spatial_shape == [4, 10, 11]
        assert np.array_equal(out.features.detach().cpu().numpy()[o], g[ykey][g["order"]])
        go = np.empty(out.features.shape, np.float32)
        go[o] = detgen.randn("pool_g", out.features.shape)[g["order"]]
        out.features.backward(T(go, dev))
        assert np.array_equal(x.grad.cpu().numpy(), g[gkey])
    # function-level entry points on a reference-format rulebook (ops.py:161-183)
    outids, pairs, num, _ = orc.get_indice_pairs(ind, CONV_BWD_BATCH, CONV_BWD_SHAPE, ks, st, pd, [1, 1, 1], 0)
    y = spconv.ops.indice_maxpool(T(f, dev), T(pairs, dev), T(num, dev), len(outids))
    assert np.array_equal(y.cpu().numpy(), orc.indice_maxpool(f, pairs, num, len(outids)))
    go = detgen.randn("pool_g2", tuple(y.shape))
    gin = spconv.ops.indice_maxpool_backward(T(f, dev), y, T(go, dev), T(pairs, dev), T(num, dev))
    assert np.array_equal(gin.cpu().numpy(), orc.indice_maxpool_backward(f, y.cpu().numpy(), go, pairs, num))


def test_sparse_maxpool_full_size_vs_oracle(dev):
    from dualfusion import ops, synth
    pts = torch.from_numpy(synth.nusc_sweep(seed=3)).to(dev)
    _, coors, _, mean = ops.hard_voxelize(pts, synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, 120000, want_voxels=False, batch_index=0)
    from dualfusion import spconv
    feats = torch.cat([mean, mean[:, :3]], 1).contiguous()                  # 8 channels
    x = spconv.SparseConvTensor(feats, coors, [41, 1440, 1440], 1)
    out = spconv.SparseMaxPool3d(3, 2, 1)(x)
    outids, pairs, num, _ = orc.get_indice_pairs(coors.cpu().numpy(), 1, [41, 1440, 1440], [3] * 3, [2] * 3, [1] * 3, [1] * 3, 0)
    want = orc.indice_maxpool(feats.cpu().numpy(), pairs, num, len(outids))
    a, b = _canon(out.indices.cpu().numpy()), _canon(outids)
    assert np.array_equal(out.indices.cpu().numpy()[a], outids[b])
    assert np.array_equal(out.features.cpu().numpy()[a], want[b])


def test_sparse_inverse_conv_golden_and_autograd(golden, dev):
    """SparseConv3d(indice_key) -> SparseInverseConv3d(same key): the inverse reads the registered rulebook backwards
    (conv.py:146-151,188-192); values against indice_conv_fp32(inverse=1) of the reference's compiled CPU code."""
    from dualfusion import spconv
    from make_golden import CONV_BWD_BATCH, CONV_BWD_SHAPE, conv_bwd_case
    g = golden("pool.npz")
    ind, ks, st, pd, f, w = conv_bwd_case(0)
    down = spconv.SparseConv3d(12, 20, 3, 2, 1, bias=False, indice_key="d").to(dev)
    up = spconv.SparseInverseConv3d(20, 12, 3, indice_key="d", bias=False).to(dev)
    wi = detgen.randn("inv_w", (3, 3, 3, 20, 12), 0.2)
    with torch.no_grad():
        down.weight.copy_(T(w, dev))
        up.weight.copy_(T(wi, dev))
        x = spconv.SparseConvTensor(T(f, dev), T(ind, dev), CONV_BWD_SHAPE, CONV_BWD_BATCH)
        mid = down(x)
        o = _canon(mid.indices.cpu().numpy())
        fo = np.empty((len(o), 20), np.float32)
        fo[o] = detgen.randn("inv_f", (len(o), 20))[g["order"]]
        y = up(mid.replace_feature(T(fo, dev)))
    assert torch.equal(y.indices, x.indices) and y.spatial_shape == CONV_BWD_SHAPE
    assert np.abs(y.features.cpu().numpy() - g["inv"]).max() <= 2e-5 * np.abs(g["inv"]).max()
    with pytest.raises(Exception):
        spconv.SparseInverseConv3d(20, 12, 3, indice_key="missing", bias=False).to(dev)(mid)
    # training path: gradients flow through both convolutions
    xg = spconv.SparseConvTensor(T(f, dev).requires_grad_(True), T(ind, dev), CONV_BWD_SHAPE, CONV_BWD_BATCH)
    up(down(xg)).features.square().sum().backward()
    assert xg.features.grad.abs().sum() > 0 and up.weight.grad.abs().sum() > 0 and down.weight.grad.abs().sum() > 0


def test_dynamic_voxelize_golden_and_oracle(golden, dev):
    from dualfusion import ops, synth, voxel
    from make_golden import POOL_RANGE, POOL_VS, pool_points
    g = golden("pool.npz")
    pts = pool_points()
    got = voxel.voxelization(T(pts, dev), POOL_VS, POOL_RANGE, -1, 20000)
    assert got.dtype == torch.int32 and np.array_equal(got.cpu().numpy(), g["dyn"])
    sweep = synth.nusc_sweep(seed=1)
    sweep[:5, 0] = [np.nan, np.inf, -np.inf, 53.999996, -54.0]
    want = orc.dynamic_voxelize(sweep, synth.NUSC_VOXEL, synth.NUSC_RANGE)
    got = ops.dynamic_voxelize(T(sweep, dev), synth.NUSC_VOXEL, synth.NUSC_RANGE).cpu().numpy()
    assert np.array_equal(got, want) and (got[:3] == -1).all()
    assert ops.dynamic_voxelize(torch.zeros((0, 4), device=dev), POOL_VS, POOL_RANGE).shape == (0, 3)


def test_sparse_conv_transpose_golden(golden, dev):
    """SparseConvTranspose3d / get_indice_pairs(transpose=True) against the reference's compiled CPU code
    (tests/golden/conv_transpose.npz): output sets bit for bit (sorted = canonical order), per-offset pair counts,
    features <= 1e-3 (measured ~1e-6); the reference-format rulebook through spconv.ops too; autograd runs."""
    from dualfusion import spconv
    from make_golden import CONV_BWD_BATCH, CONV_BWD_SHAPE, CONVT_CASES, convt_case
    g = golden("conv_transpose.npz")
    ind, f = convt_case()
    for tag, ks, st, pd, op in CONVT_CASES:
        m = spconv.SparseConvTranspose3d(16, 16, ks, stride=st, padding=pd, bias=False).to(dev).eval()
        m.output_padding = list(op)
        w = detgen.randn("convt_w_" + tag, tuple(ks) + (16, 16), 0.2)
        with torch.no_grad():
            m.weight.copy_(T(w, dev))
        x = spconv.SparseConvTensor(T(f, dev), T(ind, dev), CONV_BWD_SHAPE, CONV_BWD_BATCH)
        with torch.no_grad():
            y = m(x)
        assert list(y.spatial_shape) == list(g["oshape_" + tag])
        assert np.array_equal(y.indices.cpu().numpy(), g["outids_" + tag])
        ref_y = g["y_" + tag]
        assert np.abs(y.features.cpu().numpy() - ref_y).max() <= 1e-3 * np.abs(ref_y).max()
        outids, pairs, num = spconv.ops.get_indice_pairs(T(ind, dev), CONV_BWD_BATCH, CONV_BWD_SHAPE, ks, st, pd, 1, list(op),
                                                         subm=False, transpose=True)
        assert np.array_equal(outids.cpu().numpy(), g["outids_" + tag]) and np.array_equal(num.cpu().numpy(), g["num_" + tag])
        y2 = spconv.ops.indice_conv(T(f, dev), T(w, dev), pairs, num, outids.shape[0])
        assert np.abs(y2.cpu().numpy() - ref_y).max() <= 1e-3 * np.abs(ref_y).max()
    m.train()
    xf = T(f, dev).requires_grad_(True)
    out = m(spconv.SparseConvTensor(xf, T(ind, dev), CONV_BWD_SHAPE, CONV_BWD_BATCH))
    out.features.square().sum().backward()
    assert torch.isfinite(xf.grad).all() and float(xf.grad.abs().sum()) > 0 and float(m.weight.grad.abs().sum()) > 0


def test_sparse_conv2d_and_pool2d_vs_dense_torch(dev):
    """SparseConv2d / SubMConv2d / SparseMaxPool2d (TF/mmdet3d/ops/spconv/conv.py:207-260, pool.py:73-78) run on the
    3-D kernels as a one-slice volume; values against torch's dense conv2d / max_pool2d at the active sites."""
    import torch.nn.functional as F
    from dualfusion import spconv
    B, H, W, C = 2, 33, 40, 16
    rs = np.random.RandomState(5)
    flat = rs.choice(B * H * W, size=700, replace=False)
    ind = np.stack([flat // (H * W), (flat % (H * W)) // W, flat % W], 1).astype(np.int32)
    f = detgen.randn("c2d_f", (len(ind), C))
    dense = torch.zeros(B, C, H, W, device=dev)
    dense[ind[:, 0], :, ind[:, 1], ind[:, 2]] = T(f, dev)
    x = spconv.SparseConvTensor(T(f, dev), T(ind, dev), [H, W], B)
    subm = spconv.SubMConv2d(C, 32, 3, padding=1, bias=True, indice_key="s").to(dev)
    down = spconv.SparseConv2d(32, 32, 3, stride=2, padding=1, bias=False).to(dev)
    with torch.no_grad():
        y = subm(x)
        assert y.indices.shape[1] == 3 and torch.equal(y.indices, x.indices) and y.spatial_shape == [H, W]
        want = F.conv2d(dense, subm.weight.permute(3, 2, 0, 1), subm.bias, padding=1)
        got = y.dense()
        act = (dense.abs().sum(1, keepdim=True) > 0).float()
        assert (got - want * act).abs().max() <= 1e-4 * want.abs().max()
        z = down(y)
        assert z.spatial_shape == [17, 20] and z.indices.shape[1] == 3
        want2 = F.conv2d(got, down.weight.permute(3, 2, 0, 1), None, stride=2, padding=1)
        assert (z.dense() - want2).abs().max() <= 1e-4 * want2.abs().max()
        p = spconv.SparseMaxPool2d(3, 2, 1)(y)
        wantp = F.max_pool2d(got.clamp_min(0), 3, 2, 1)                    # the reference's pool starts from zero
        assert torch.equal(p.indices, z.indices) and (p.dense() - wantp).abs().max() <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,K,n_in,n_out,density", [
    (128, 128, 27, 3000, 3000, 0.4), (64, 64, 27, 5000, 4100, 0.3), (32, 64, 27, 2500, 900, 0.5),
    (16, 16, 27, 20000, 20000, 0.35), (16, 32, 8, 7000, 1777, 0.9), (64, 128, 27, 1300, 300, 0.02),
    (256, 256, 9, 2100, 2100, 0.95), (48, 20, 27, 999, 1001, 0.4), (128, 64, 1, 70, 70, 1.0), (12, 20, 27, 500, 400, 0.5)])
def test_filter_gradient_kernels_against_float64(cin, cout, K, n_in, n_out, density):
    """df3d_sparse_conv_grad_filters: the LDS-staged pair-compacted kernel (channel counts that are multiples of 4) and the
    direct kernel (everything else, DF3D_WGRAD=0) against sum over pairs of features[in]^T grad_out[out] in float64:
    ragged row counts, empty offsets, channel counts that do not fill a tile."""
    import os
    from dualfusion import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(cin * 1000 + cout)
    nbr = torch.randint(0, n_in, (K, n_out), generator=gen, dtype=torch.int32)
    nbr[torch.rand((K, n_out), generator=gen) >= density] = -1
    if K > 2:
        nbr[1] = -1                                        # an offset without pairs
    feats = torch.randn((n_in, cin), generator=gen)
    gout = torch.randn((n_out, cout), generator=gen)
    ref = torch.zeros((K, cin, cout), dtype=torch.float64)
    for k in range(K):
        o = (nbr[k] >= 0).nonzero(as_tuple=True)[0]
        ref[k] = feats[nbr[k][o].long()].double().t() @ gout[o].double()
    scale = max(1.0, float(ref.abs().max()))
    outs = {}
    for mode in ("1", "0", "3"):
        if mode == "0" and cout > 128:
            continue
        if mode == "3" and (cin % 4 or cout % 4):
            continue
        os.environ["DF3D_WGRAD"] = mode
        try:
            outs[mode] = ops.sparse_conv_grad_filters(feats.to(dev), gout.to(dev), nbr.to(dev)).cpu().double()
        finally:
            os.environ.pop("DF3D_WGRAD", None)
        # (round 5) mode 3 = three bf16 parts per operand on the 16-bit matrix cores, forced here on every shape with channel
        # counts that are multiples of 4 (by default it takes the layers with >= 64 channels on both sides): fp32-grade
        assert float((outs[mode] - ref).abs().max()) <= (4e-6 if mode == "3" else 2e-5) * scale, mode
    if cin % 4 == 0 and cout % 4 == 0 and cin >= 32 and cout >= 32:
        # second half of round 5: fp16 pairs, the gradient under its power-of-two block scale (three products)
        for gs in (1.0, 1e-7, 1e+6):
            gd = (gout * gs).to(dev)
            got = ops.sparse_conv_grad_filters(feats.to(dev), gd, nbr.to(dev), grad_scale=ops.rows_pow2_scale(gd)).cpu().double()
            assert float((got - ref * gs).abs().max()) <= 4e-6 * scale * gs, gs
        # round 6: bf16 mixed-precision training -- ONE bf16 part per operand, one product: exact (fp32 accumulate) for operands
        # that are bf16 values already, bf16-grade (2^-8 per operand, averaging over the pairs) for fp32 operands
        fb, gb = feats.bfloat16().float(), gout.bfloat16().float()
        refb = torch.zeros((K, cin, cout), dtype=torch.float64)
        for k in range(K):
            o = (nbr[k] >= 0).nonzero(as_tuple=True)[0]
            refb[k] = fb[nbr[k][o].long()].double().t() @ gb[o].double()
        got = ops.sparse_conv_grad_filters(fb.to(dev), gb.to(dev), nbr.to(dev), bf16=True).cpu().double()
        assert float((got - refb).abs().max()) <= 4e-6 * scale
        got = ops.sparse_conv_grad_filters(feats.to(dev), gout.to(dev), nbr.to(dev), bf16=True).cpu().double()
        if cin >= 64 and cout >= 64:                                           # (narrower layers stay on the fp32 kernel)
            assert float((got - refb).abs().max()) <= 4e-6 * scale            # round-to-nearest-even, as torch's .bfloat16()
        assert float((got - ref).abs().max()) <= 1e-2 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("n,cin,cout,gscale", [(32034, 128, 1024, 1e-6), (31254, 1024, 128, 3.0), (5000, 256, 128, 1e-3),
                                               (70, 64, 64, 1.0), (1, 128, 128, 1.0), (4099, 132, 64, 1e-9), (40050, 4, 256, 1.0)])
def test_rows_grad_weights_against_float64(n, cin, cout, gscale):
    """df3d_rows_grad_weights (the filter-gradient kernel without a table) = x^T grad over rows, the weight gradient of the
    adapter's linear layers: against float64 at the feed-forward / projection shapes of the training step, with gradients of
    very different magnitudes (the three-part bf16 split has fp32's exponent range: no scaling anywhere) and a row block whose
    gradient is 2^40 times larger than the rest."""
    from dualfusion import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(n + cin)
    x = torch.randn((n, cin), generator=gen) * 2.0
    g = torch.randn((n, cout), generator=gen) * gscale
    if n > 1000:
        g[100:132] *= 2.0 ** 40
    ref = x.double().t() @ g.double()
    got = ops.rows_grad_weights(x.to(dev), g.to(dev)).cpu().double()
    assert float((got - ref).abs().max()) <= 4e-6 * float(ref.abs().max())
    if n <= 1000 or gscale == 1.0 or True:
        # two-part form: the gradient operand under its block scale (rows a factor 2^40 apart exceed 22 bits of ONE scale: the
        # small rows then count for nothing against the large ones, in float64 as well -- the bound is of the result's scale)
        gd = g.to(dev)
        got2 = ops.rows_grad_weights(x.to(dev), gd, g_scale=ops.rows_pow2_scale(gd)).cpu().double()
        assert float((got2 - ref).abs().max()) <= 4e-6 * float(ref.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("gscale", [1.0, 1e-9, 3e+7])
def test_input_gradient_on_block_scaled_two_part_rows(gscale):
    """ops.sparse_conv_backward in the "split" mode: the gradient rows are split into fp16 pairs under their own power-of-two
    scale (df3d_split_rows_scaled) and the convolution's epilogue undoes it -- against float64, for gradient tensors of very
    different magnitudes (the fixed-scale split would underflow at 1e-9 and overflow at 3e+7), with entries spread over twelve
    decades inside one tensor, and against the three-bf16-part path (DF3D_GRAD_SCALED=0); the range flag stays clear."""
    import os
    from dualfusion import ops
    if ops.CONV_PRECISION != "split":
        pytest.skip("the block scale replaces the three-part rows of the split mode")
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    K, cin, cout, n = 27, 64, 128, 6000
    # (a rulebook pairs an input with at most one output per offset: a permutation per offset, thinned)
    nbr = torch.stack([torch.randperm(n, generator=gen) for _ in range(K)]).to(torch.int32)
    nbr[torch.rand((K, n), generator=gen) >= 0.4] = -1
    feats = torch.randn((n, cin), generator=gen)
    filt = torch.randn((K, cin, cout), generator=gen) * 0.1
    g = torch.randn((n, cout), generator=gen) * gscale
    g[::3] *= 1e-6                                               # a third of the rows a million times smaller
    g[1::7] *= 1e-12
    inv = ops.invert_neighbors(nbr.to(dev), n)
    ref = torch.zeros((n, cin), dtype=torch.float64)
    for k in range(K):
        o = (nbr[k] >= 0).nonzero(as_tuple=True)[0]
        ref.index_add_(0, nbr[k][o].long(), g[o].double() @ filt[k].double().t())
    got, _ = ops.sparse_conv_backward(feats.to(dev), filt.to(dev), g.to(dev), nbr.to(dev), subm=False, inv=inv)
    scale = float(ref.abs().max())
    assert float((got.cpu().double() - ref).abs().max()) <= 4e-6 * scale
    old = os.environ.get("DF3D_GRAD_SCALED")
    os.environ["DF3D_GRAD_SCALED"] = "0"
    try:
        three, _ = ops.sparse_conv_backward(feats.to(dev), filt.to(dev), g.to(dev), nbr.to(dev), subm=False, inv=inv)
    finally:
        if old is None:
            os.environ.pop("DF3D_GRAD_SCALED", None)
        else:
            os.environ["DF3D_GRAD_SCALED"] = old
    assert float((three.cpu().double() - ref).abs().max()) <= 4e-6 * scale
    assert not ops.split_overflow(reset=True)[0]
    # an all-zero gradient and one with an infinity: scale 1; the infinity is reported, not swallowed
    z, _ = ops.sparse_conv_backward(feats.to(dev), filt.to(dev), torch.zeros_like(g).to(dev), nbr.to(dev), subm=False, inv=inv)
    assert float(z.abs().max()) == 0.0
    g2 = g.clone()
    g2[5, 3] = float("inf")
    ops.sparse_conv_backward(feats.to(dev), filt.to(dev), g2.to(dev), nbr.to(dev), subm=False, inv=inv)
    assert ops.split_overflow(reset=True)[0]


@pytest.mark.gpu
@pytest.mark.parametrize("N,Lq,M,D,P,H,W", [(3, 700, 8, 16, 4, 37, 61), (2, 300, 4, 16, 4, 8, 8), (1, 9000, 8, 16, 4, 112, 200),
                                            (2, 64, 2, 16, 2, 5, 19), (1, 40, 1, 16, 16, 3, 2)])
def test_msda_backward_binned_against_the_atomic_kernel_and_the_oracle(N, Lq, M, D, P, H, W):
    """Round 6: the value gradient without global atomics (df3d_ms_deform_attn_backward_binned: sampling points counting-sorted
    by (8 x 8 pixel tile, head), a 9 x 9 pixel footprint per wave as a matrix product on the fp32 MFMA, slabs gathered) against
    the one-atomic-per-contribution kernel and the oracle's col2im: maps that are not multiples of the tile, points outside / on
    the border, a third of the queries on one pixel (the unseen voxels' reference point: several work items per bin at the
    largest case), rows without an upstream gradient (the padded rows)."""
    import os
    from dualfusion import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(N * 1000 + Lq)
    value = torch.randn(N, H * W, M, D, generator=gen)
    ref = torch.rand(N, Lq, 1, 1, 1, 2, generator=gen) * 1.2 - 0.1
    ref[:, Lq * 2 // 3:] = 0.0
    loc = (ref + torch.randn(N, Lq, M, 1, P, 2, generator=gen) * 0.03).contiguous()
    aw = torch.softmax(torch.randn(N, Lq, M, P, generator=gen), -1).view(N, Lq, M, 1, P).contiguous()
    go = torch.randn(N, Lq, M * D, generator=gen)
    go[:, Lq * 5 // 6:] = 0.0
    shp = torch.tensor([[H, W]], dtype=torch.long, device=dev)
    ls = torch.zeros(1, dtype=torch.long, device=dev)
    args = [t.to(dev) for t in (value,)] + [shp, ls] + [t.to(dev) for t in (loc, aw, go)]
    from dualfusion import _lib
    real = _lib.load().df3d_ms_deform_attn_backward_binned
    got = [t.cpu() for t in ops.ms_deform_attn_backward(*args)]
    again = [t.cpu() for t in ops.ms_deform_attn_backward(*args)]
    assert torch.equal(got[1], again[1]) and torch.equal(got[2], again[2])
    assert float((got[0] - again[0]).abs().max()) <= 2e-5 * max(1.0, float(got[0].abs().max()))   # (order of the points in a bin)
    assert int(_lib.load().df3d_ms_deform_attn_backward_binned_slab_bytes(N, M, D, Lq, P, H, W)) > 0 and real is not None
    os.environ["DF3D_MSDA_BWD"] = "atomic"
    try:
        old = [t.cpu() for t in ops.ms_deform_attn_backward(*args)]
    finally:
        os.environ.pop("DF3D_MSDA_BWD", None)
    want = orc.ms_deform_attn_backward(value.numpy(), [(H, W)], loc.numpy(), aw.numpy(), go.numpy())
    for a, b, w, name in zip(got, old, want, ("value", "loc", "weight")):
        scale = max(1.0, float(np.abs(w).max()))
        assert float((a - b).abs().max()) <= 2e-5 * scale, name                   # (summation order of the atomics)
        assert np.abs(a.numpy() - w).max() <= 1e-4 * scale, name
    assert torch.equal(got[1], old[1]) and torch.equal(got[2], old[2])            # the gather half is the same code


@pytest.mark.gpu
def test_linear_module_under_autocast_trains_like_nn_linear():
    """`dualfusion.linear_rows.Linear` replaces nn.Linear throughout ACTR / MSDeformAttn, and the reference trains
    those layers with AMP.  Under torch.autocast the module must behave like nn.Linear (half-precision forward, gradients in
    the parameters' dtype) instead of handing half-precision gradients to the fp32 row kernels."""
    from dualfusion.linear_rows import Linear
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    x = torch.randn((4, 3000, 128), generator=gen).to(dev).requires_grad_(True)
    lin = Linear(128, 256).to(dev)
    ref = torch.nn.Linear(128, 256).to(dev)
    ref.load_state_dict(lin.state_dict())
    xr = x.detach().clone().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y = lin(x)
        yr = ref(xr)
    assert y.dtype == yr.dtype == torch.bfloat16
    y.float().square().sum().backward()
    yr.float().square().sum().backward()
    for a, b in ((lin.weight.grad, ref.weight.grad), (lin.bias.grad, ref.bias.grad), (x.grad, xr.grad)):
        assert a.dtype == torch.float32 and torch.equal(a, b)
    # a caller that casts the OUTPUT of the fp32 path itself: half-precision gradients reach the Function and are accepted
    lin.zero_grad()
    y2 = lin(x.detach()).to(torch.bfloat16)
    y2.float().sum().backward()
    assert lin.weight.grad is not None and bool(torch.isfinite(lin.weight.grad).all())


@pytest.mark.gpu
def test_linear_module_and_gate_weight_gradients_on_native_kernels():
    """dualfusion.linear_rows.Linear (forward = F.linear; weight gradient on df3d_rows_grad_weights) and the one-output
    channel-first gate (`ops.channel_first_linear`, weight gradient on df3d_chanfirst_dot) against torch's autograd in float64:
    3-D inputs, a bias, an output count that is not a multiple of 4, a pixel count that is not a multiple of 4."""
    from dualfusion import ops
    from dualfusion.linear_rows import Linear
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(11)
    for shape, cout in (((6, 5339, 128), 1024), ((3, 4000, 256), 131), ((40050, 128), 1)):
        x = torch.randn(shape, generator=gen)
        lin = Linear(shape[-1], cout)
        ref = torch.nn.Linear(shape[-1], cout).double()
        ref.load_state_dict({k: v.double() for k, v in lin.state_dict().items()})
        g = torch.randn(shape[:-1] + (cout,), generator=gen) * 1e-4
        xr = x.double().requires_grad_(True)
        ref(xr).backward(g.double())
        lin = lin.to(dev)
        xd = x.to(dev).requires_grad_(True)
        y = lin(xd)
        assert type(y.grad_fn).__name__ == "_LinearFunctionBackward"
        y.backward(g.to(dev))
        for got, want in ((lin.weight.grad, ref.weight.grad), (lin.bias.grad, ref.bias.grad), (xd.grad, xr.grad)):
            assert float((got.cpu().double() - want).abs().max()) <= 4e-6 * float(want.abs().max())
    for S in (40050, 40051):
        x = torch.randn((6, 256, S), generator=gen)
        w = torch.randn((1, 256), generator=gen)
        g = torch.randn((6, 1, S), generator=gen) * 1e-3
        wr = w.double().requires_grad_(True)
        torch.matmul(wr, x.double()).backward(g.double())
        wd = w.to(dev).requires_grad_(True)
        ops.channel_first_linear(x.to(dev), wd).backward(g.to(dev))
        assert float((wd.grad.cpu().double() - wr.grad).abs().max()) <= 4e-6 * float(wr.grad.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("n,c,relu", [(37001, 16, True), (9000, 32, False), (5003, 64, True), (3000, 128, True),
                                      (2048, 256, False), (777, 512, True), (400, 2304, True), (2, 64, False)])
def test_batch_norm_rows_kernels_vs_float64(n, c, relu):
    """df3d_bn_rows_forward / _backward through `ops.batch_norm_rows` against torch's BatchNorm (+ ReLU) in float64: output,
    input / weight / bias gradients, running statistics, the batch counter."""
    from dualfusion import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(n + c)
    x = torch.randn((n, c), generator=gen) * 2.0 + torch.randn((1, c), generator=gen)
    g = torch.randn((n, c), generator=gen)
    ref = torch.nn.BatchNorm1d(c, eps=1e-3, momentum=0.01).double().train()
    with torch.no_grad():
        ref.weight.copy_(torch.randn(c, generator=gen).double())
        ref.bias.copy_(torch.randn(c, generator=gen).double())
        ref.running_mean.copy_(torch.randn(c, generator=gen).double())
        ref.running_var.copy_(torch.rand(c, generator=gen).double() + 0.5)
    import copy
    bn = copy.deepcopy(ref).float().to(dev)
    assert ops.bn_rows_supported(c)
    xr = x.double().requires_grad_(True)
    yr = ref(xr)
    yr = torch.relu(yr) if relu else yr
    (yr * g.double()).sum().backward()
    xd = x.to(dev).requires_grad_(True)
    yd = ops.batch_norm_rows(bn, xd, relu)
    (yd * g.to(dev)).sum().backward()
    rel = lambda a, b: float((a.detach().cpu().double() - b).abs().max() / max(1e-9, float(b.abs().max())))
    tol = 1e-4 if n > 2 else 2e-3          # two rows: xhat = +-1 and the input gradient is all cancellation
    assert rel(yd, yr.detach()) < 1e-5
    assert rel(xd.grad, xr.grad) < tol
    assert rel(bn.weight.grad, ref.weight.grad) < tol and rel(bn.bias.grad, ref.bias.grad) < tol
    assert rel(bn.running_mean, ref.running_mean) < 1e-5 and rel(bn.running_var, ref.running_var) < 1e-5
    assert int(bn.num_batches_tracked) == int(ref.num_batches_tracked) == 1


@pytest.mark.gpu
def test_half_entry_points_of_the_spconv_mirror(dev):
    """`*_half` registrations of the reference extension (src/all.cc:21-51; ops.py:112-184 dispatches on the dtype): fp16
    features / filters through `spconv.ops.indice_conv`, `fused_indice_conv`, `indice_conv_backward`, `indice_maxpool` --
    fp16 in, fp16 out, equal to the fp32 entry on the widened operands rounded once."""
    from dualfusion import spconv
    shape, batch = [8, 20, 24], 2
    ind = detgen.clustered_voxels("half", batch, shape, n_seeds=4, walk=150)
    f = (detgen.randn("half_f", (len(ind), 16)) * 0.5).astype(np.float16)
    w = (detgen.randn("half_w", (3, 3, 3, 16, 32)) * 0.1).astype(np.float16)
    b = detgen.randn("half_b", (32,), 0.1).astype(np.float16)
    outids, pairs, num = spconv.ops.get_indice_pairs(T(ind, dev), batch, shape, 3, 2, 1, 1, subm=False)
    n_out = outids.shape[0]
    fh, wh, bh = T(f, dev), T(w, dev), T(b, dev)
    assert fh.dtype == torch.float16
    y = spconv.ops.indice_conv(fh, wh, pairs, num, n_out)
    y32 = spconv.ops.indice_conv(fh.float(), wh.float(), pairs, num, n_out)
    assert y.dtype == torch.float16 and torch.equal(y, y32.half())
    yb = spconv.ops.fused_indice_conv(fh, wh, bh, pairs, num, n_out, False, False)
    assert yb.dtype == torch.float16
    assert torch.equal(yb, spconv.ops.fused_indice_conv(fh.float(), wh.float(), bh.float(), pairs, num, n_out, False, False).half())
    go = T((detgen.randn("half_g", (n_out, 32)) * 0.5).astype(np.float16), dev)
    gi, gw = spconv.ops.indice_conv_backward(fh, wh, go, pairs, num)
    gi32, gw32 = spconv.ops.indice_conv_backward(fh.float(), wh.float(), go.float(), pairs, num)
    assert gi.dtype == gw.dtype == torch.float16 and tuple(gw.shape) == tuple(wh.shape)
    assert torch.equal(gi, gi32.half())
    assert float((gw.float() - gw32).abs().max()) <= 2e-3 * float(gw32.abs().max())      # atomics: summation order varies
    p = spconv.ops.indice_maxpool(fh, pairs, num, n_out)
    assert p.dtype == torch.float16 and torch.equal(p, spconv.ops.indice_maxpool(fh.float(), pairs, num, n_out).half())
    # float64 reference of the convolution itself (fp16 operands are exact in float64)
    pr, nm = pairs.cpu().numpy(), num.cpu().numpy()
    ref = np.zeros((n_out, 32))
    f64, w64 = f.astype(np.float64), w.astype(np.float64).reshape(27, 16, 32)
    for k in range(27):
        i, o = pr[k, 0, :nm[k]], pr[k, 1, :nm[k]]
        np.add.at(ref, o, f64[i] @ w64[k])
    assert np.abs(y.float().cpu().numpy() - ref).max() <= 2e-3 * np.abs(ref).max()


@pytest.mark.gpu
@pytest.mark.parametrize("N,m,ns,rmin,rmax", [(24000, 2051, 16, 0.0, 1.0), (5000, 300, 70, 0.5, 3.0), (1024, 16, 8, 0.0, 0.05),
                                              (777, 5, 32, 1.0, 40.0)])
def test_ball_query_wave_kernel_vs_oracle(dev, N, m, ns, rmin, rmax):
    """The wave-per-centre ball query against the reference's serial scan (oracle): index order of the hits, the first hit
    repeated in unused slots, zero rows for centres without a hit, the d2 == 0 clause with a positive inner radius, more
    samples than lanes, row counts that do not fill a tile / a workgroup."""
    from dualfusion import ops
    xyz = detgen.rand("bqw_xyz%d" % N, (2, N, 3), -20, 20)
    new_xyz = np.concatenate([xyz[:, : m // 2], detgen.rand("bqw_c%d" % N, (2, m - m // 2, 3), -25, 25)], 1)
    got = ops.ball_query(rmin, rmax, ns, T(xyz, dev), T(new_xyz, dev)).cpu().numpy()
    want = orc.ball_query(rmin, rmax, ns, xyz, new_xyz)
    assert np.array_equal(got, want)
    assert (want[:, m // 2:].max(-1) == 0).any() or rmax > 10          # some centres really have no hit


@pytest.mark.gpu
@pytest.mark.parametrize("rows,C,n_feat", [(5000, 64, 900), (333, 32, 50), (70, 128, 7), (1, 16, 3)])
def test_pe_gather_add_vs_torch(dev, rows, C, n_feat):
    """df3d_pe_gather_add = feat[sel] + W1 relu(W0 xyz + b0) + b1 against the torch expression in float64."""
    from dualfusion import ops
    gen = torch.Generator().manual_seed(rows + C)
    feat = torch.randn((n_feat, C), generator=gen)
    sel = torch.randint(0, n_feat, (rows,), generator=gen)
    xyz = torch.randn((rows, 3), generator=gen) * 3
    w0, b0 = torch.randn((C // 2, 3), generator=gen), torch.randn((C // 2,), generator=gen)
    w1, b1 = torch.randn((C, C // 2), generator=gen) * 0.3, torch.randn((C,), generator=gen)
    want = feat.double()[sel] + torch.relu(xyz.double() @ w0.double().t() + b0.double()) @ w1.double().t() + b1.double()
    got = ops.pe_gather_add(*[t.to(dev) for t in (feat, sel, xyz, w0, b0, w1, b1)]).cpu().double()
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())



@pytest.mark.gpu
def test_round2_entries_on_empty_and_minimal_inputs(dev):
    """Empty / one-row inputs of the entries added in round 2: no launch, no fault, the reference's result shape."""
    from dualfusion import ops
    # filter gradient: no output rows, no input rows, a table without a single pair
    nbr0 = torch.empty((27, 0), dtype=torch.int32, device=dev)
    gw = ops.sparse_conv_grad_filters(torch.randn(5, 16, device=dev), torch.empty((0, 32), device=dev), nbr0)
    assert tuple(gw.shape) == (27, 16, 32) and float(gw.abs().max()) == 0.0
    nbr = torch.full((27, 9), -1, dtype=torch.int32, device=dev)
    gw = ops.sparse_conv_grad_filters(torch.randn(5, 16, device=dev), torch.randn(9, 32, device=dev), nbr)
    assert float(gw.abs().max()) == 0.0
    nbr[13, 4] = 2                                             # exactly one pair
    f, g = torch.randn(5, 16, device=dev), torch.randn(9, 32, device=dev)
    gw = ops.sparse_conv_grad_filters(f, g, nbr)
    torch.testing.assert_close(gw[13], torch.outer(f[2], g[4]), rtol=1e-6, atol=1e-6)
    assert float(gw[:13].abs().max()) == 0.0 and float(gw[14:].abs().max()) == 0.0
    # group attention / positional gather / ball query without groups, rows, centres
    assert tuple(ops.group_attention(torch.empty((0, 192), device=dev), 32, 0, 4).shape) == (0, 64)
    o = ops.group_attention(torch.randn(1, 192, device=dev), 1, 1, 4)              # one token: softmax of one score = v
    assert tuple(o.shape) == (1, 64)
    e = ops.pe_gather_add(torch.randn(3, 64, device=dev), torch.empty((0,), dtype=torch.int64, device=dev),
                          torch.empty((0, 3), device=dev), torch.randn(32, 3, device=dev), torch.randn(32, device=dev),
                          torch.randn(64, 32, device=dev), torch.randn(64, device=dev))
    assert tuple(e.shape) == (0, 64)
    bq = ops.ball_query(0.0, 1.0, 8, torch.randn(2, 50, 3, device=dev), torch.empty((2, 0, 3), device=dev))
    assert tuple(bq.shape) == (2, 0, 8)
    bq = ops.ball_query(0.0, 1e-6, 8, torch.ones(1, 3, 3, device=dev), torch.zeros(1, 2, 3, device=dev))   # no hit at all
    assert int(bq.abs().max()) == 0
    # BatchNorm rows: a single row takes the module's own path (no batch statistics to speak of), two rows the kernels
    bn = torch.nn.BatchNorm1d(16).to(dev).train()
    y = ops.batch_norm_rows(bn, torch.randn(2, 16, device=dev), relu=True)
    assert tuple(y.shape) == (2, 16) and bool(torch.isfinite(y).all()) and int(bn.num_batches_tracked) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cin,cout,bias", [(20000, 64, 64, True), (16500, 128, 64, False), (17000, 64, 128, True),
                                                (100, 64, 64, True), (20000, 48, 64, True)])
def test_linear_rows_helper(dev, rows, cin, cout, bias):
    """`ops.linear_rows` = nn.Linear over many narrow rows on the split-precision conv kernel (identity table) when the
    shape is served, torch otherwise (few rows, unsupported widths, grad mode): values against float64 either way."""
    from dualfusion import ops
    gen = torch.Generator().manual_seed(rows + cin)
    lin = torch.nn.Linear(cin, cout, bias=bias)
    x = torch.randn((2, rows // 2, cin), generator=gen)
    want = torch.nn.functional.linear(x.double(), lin.weight.double(), lin.bias.double() if bias else None)
    lin = lin.to(dev)
    with torch.no_grad():
        got = ops.linear_rows(x.to(dev), lin)
    assert tuple(got.shape) == (2, rows // 2, cout)
    assert float((got.cpu().double() - want).abs().max()) <= 1e-4 * float(want.abs().max())
    y = ops.linear_rows(x.to(dev).requires_grad_(True), lin)               # grad mode: the module itself
    assert y.requires_grad


@pytest.mark.gpu
def test_split_row_outputs_of_layernorm_and_group_attention(dev):
    """The split rows `add_layernorm(want_split=True)` and `group_attention(split_only=True)` write equal `split_rows` of
    their fp32 outputs bit for bit."""
    from dualfusion import ops
    gen = torch.Generator().manual_seed(11)
    for C in (64, 128, 256):
        x, y = torch.randn((777, C), generator=gen).to(dev), torch.randn((777, C), generator=gen).to(dev)
        w, b = torch.randn(C, generator=gen).to(dev), torch.randn(C, generator=gen).to(dev)
        out, sp = ops.add_layernorm(x, y, w, b, 1e-5, want_split=True)
        assert torch.equal(out, ops.add_layernorm(x, y, w, b, 1e-5)) and torch.equal(sp, ops.split_rows(out))
    qkv = torch.randn((32 * 41, 192), generator=gen).to(dev)
    o = ops.group_attention(qkv, 32, 41, 4)
    assert torch.equal(ops.group_attention(qkv, 32, 41, 4, split_only=True), ops.split_rows(o))


@pytest.mark.parametrize("cin,couts,rows", [(128, (64, 32), 18611), (128, (128,), 777), (256, (128,), 4097), (128, (32,), 33), (128, (16, 16), 1)])
def test_rows_linear_vs_float64(dev, cin, couts, rows):
    """df3d_rows_linear (csrc/rowlinear.hip): mixed operands formed on load, two output tensors, the LayerNorm epilogue --
    against float64 (fp32-grade: <= 2e-5 of the output scale)."""
    from dualfusion import ops
    gen = torch.Generator().manual_seed(rows)
    lins = [torch.nn.Linear(cin, c) for c in couts]
    for l in lins:
        l.weight.data = torch.randn(l.weight.shape, generator=gen) / np.sqrt(cin)
        l.bias.data = torch.randn(l.bias.shape, generator=gen)
        l.to(dev)
    x0, x1, x2 = (torch.randn(3, rows // 3 + 1, cin, generator=gen).to(dev)[:, :max(rows // 3, 1)] .contiguous() for _ in range(3))
    pk = ops.rows_linear_pack([l.weight for l in lins], [l.bias for l in lins])
    W = torch.cat([l.weight for l in lins]).double().cpu()
    b = torch.cat([l.bias for l in lins]).double().cpu()
    a0 = (x0 + x2).double().cpu()
    a1 = ((x0 + x2) + (x1 + x2)).double().cpu()

    def close(got, want):
        assert got.shape == want.shape
        assert float((got.double().cpu() - want).abs().max()) <= 4e-6 * max(float(want.abs().max()), 1e-6)
    if len(couts) == 2:
        o0, o1 = ops.rows_linear(x0, pk, x1=x1, x2=x2, csplit=couts[0], n0=couts[0], n1=couts[1])
        close(o0, a0 @ W[:couts[0]].t() + b[:couts[0]])
        close(o1, a1 @ W[couts[0]:].t() + b[couts[0]:])
    else:
        close(ops.rows_linear(x0, pk), x0.double().cpu() @ W.t() + b)
        close(ops.rows_linear(x0, pk, x2=x2), a0 @ W.t() + b)
        if couts[0] == 128 and cin == 128:
            norm = torch.nn.LayerNorm(128).to(dev)
            norm.weight.data.uniform_(0.5, 1.5)
            norm.bias.data.normal_()
            res = torch.randn(x0.shape, generator=gen).to(dev)
            got = ops.rows_linear(x0, pk, ln=(res, norm))
            y = res.double().cpu() + x0.double().cpu() @ W.t() + b
            want = torch.nn.functional.layer_norm(y, (128,), norm.weight.double().cpu(), norm.bias.double().cpu(), norm.eps)
            close(got, want)
    with pytest.raises(Exception):
        ops.rows_linear_pack(torch.zeros(200, 128, device=dev))                 # > 128 outputs
    with pytest.raises(Exception):
        ops.rows_linear(x0.cpu(), pk)


@pytest.mark.parametrize("cin,ld,c0,cout", [(64, 128, 64, 16), (64, 128, 0, 128), (32, 96, 32, 64), (256, 512, 256, 256),
                                            (512, 512, 0, 256)])
def test_sparse_conv_fused_on_column_slices(dev, cin, ld, c0, cout):
    """df3d_sparse_conv_fused_ld: the exact-fp32 MFMA kernel reading a column slice of wider rows in place == the same
    convolution of a contiguous copy of the slice (bit-exact: same kernel, same order) and the float64 gather-GEMM."""
    from dualfusion import ops
    DEV = dev
    g = torch.Generator().manual_seed(cin * 7 + cout)
    n, K = 3000, 9
    wide = torch.randn(n, ld, generator=g).to(DEV)
    w = (torch.randn(K, cin, cout, generator=g) * 0.1).to(DEV)
    nbr = torch.randint(-1, n, (K, n), generator=g, dtype=torch.int32).to(DEV)
    bias = torch.randn(cout, generator=g).to(DEV)
    x = wide[:, c0:c0 + cin]
    got = ops.sparse_conv_fused(x, w, nbr, n, bias=bias, relu=True)
    if ld != cin:
        want = ops.sparse_conv_fused(x.contiguous(), w, nbr, n, bias=bias, relu=True)
        assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    xd, wd = x.double().cpu(), w.double().cpu()
    ref = torch.zeros(n, cout, dtype=torch.float64)
    nb = nbr.cpu().long()
    for k in range(K):
        m = nb[k] >= 0
        ref[m] += xd[nb[k][m]] @ wd[k]
    ref = torch.relu(ref + bias.double().cpu())
    assert (got.double().cpu() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


@pytest.mark.parametrize("G,cin,gin,cout", [(6, 64, 64, 16), (3, 64, 0, 128), (4, 32, 32, 32), (2, 128, 0, 256)])
def test_sparse_conv_grouped_vs_per_group_launches(dev, G, cin, gin, cout):
    """df3d_sparse_conv_grouped: G convolutions over one neighbour table in one launch (operands as column slices of wider
    rows) == G launches of the same kernel on contiguous copies, bit for bit; out_col places the block inside wider rows."""
    from dualfusion import ops
    g = torch.Generator().manual_seed(G * 131 + cin)
    n, K = 70000 if cout == 16 else 5000, 9
    wide = torch.randn(n, max(cin, (G - 1) * gin + cin), generator=g).to(dev)
    w = (torch.randn(G, K, cin, cout, generator=g) * 0.1).to(dev)
    nbr = torch.randint(-1, n, (K, n), generator=g, dtype=torch.int32).to(dev)
    bias, scale, shift = (torch.randn(G * cout, generator=g).to(dev) for _ in range(3))
    out = torch.full((n, 8 + G * cout + 4), -7.0, device=dev)
    got = ops.sparse_conv_grouped(wide, w, nbr, n, bias=bias, scale=scale, shift=shift, relu=True, group_in=gin, out=out, out_col=8)
    assert got is out and bool((out[:, :8] == -7).all()) and bool((out[:, 8 + G * cout:] == -7).all())
    for i in range(G):
        x = wide[:, i * gin:i * gin + cin].contiguous()
        sl = slice(i * cout, (i + 1) * cout)
        want = ops.sparse_conv_grouped(x, w[i:i + 1].contiguous(), nbr, n, bias=bias[sl].contiguous(), scale=scale[sl].contiguous(),
                                       shift=shift[sl].contiguous(), relu=True)
        assert torch.equal(out[:, 8 + i * cout:8 + (i + 1) * cout], want), i
    with pytest.raises(Exception):
        ops.sparse_conv_grouped(wide[:, :cin], w, nbr, n, group_in=max(gin, 4))      # the groups' slices leave the rows


def test_hard_voxelize_clouds_equals_per_cloud_calls(dev):
    """hard_voxelize_clouds (all clouds queued, one round trip for the counts) == the per-cloud calls it replaces, bit for bit,
    with and without work queued while waiting; a cloud without points in range contributes nothing."""
    from dualfusion import ops, synth
    clouds = [torch.from_numpy(synth.kitti_sweep(seed=30 + b)[:, :4].copy()).to(dev) for b in range(3)]
    clouds.insert(1, torch.full((50, 4), 1000.0, device=dev))                       # nothing inside the range
    feats, coors = [], []
    for b, pts in enumerate(clouds):
        _, c, _, mean = ops.hard_voxelize(pts, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 40000, want_voxels=False, batch_index=b)
        feats.append(mean)
        coors.append(c)
    flag = []
    for ww in (None, lambda: flag.append(1)):
        f, c = ops.hard_voxelize_clouds(clouds, synth.KITTI_VOXEL, synth.KITTI_RANGE, 5, 40000, while_waiting=ww)
        assert torch.equal(f, torch.cat(feats)) and torch.equal(c, torch.cat(coors))
    assert flag == [1] and not bool((torch.cat(coors)[:, 0] == 1).any())


def test_assemble_queries_by_slot_equals_by_voxel(dev):
    """df3d_assemble_queries2 with the lanes over the queries of an image (taken for >= 12 images when the list lengths are
    given: B = 4 x 3 cameras here) against the wave-per-voxel kernel (no list lengths):
    identical query tensors, incl. gate, depth position embedding, images given as a pointer table, empty lists."""
    import ctypes
    from dualfusion import _lib, ops
    lib = _lib.load()
    g = torch.Generator().manual_seed(11)
    B, ncam, H, W, C, Ci, n = 4, 3, 20, 36, 128, 256, 5000
    ind = torch.zeros(n, 4, dtype=torch.int32)
    ind[:, 0] = (torch.arange(n) >= 2600).int() + (torch.arange(n) >= 3700).int() + (torch.arange(n) >= 4400).int()
    mask = (torch.rand(ncam, n, generator=g) < 0.3).to(torch.uint8)
    mask[2, :2600] = 0                                               # an empty list (sample 0, camera 2)
    grid = torch.stack([torch.randint(0, W, (ncam, n), generator=g), torch.randint(0, H, (ncam, n), generator=g)], 2).int()
    feat, pinv = torch.randn(n, C, generator=g), torch.rand(n, 3, generator=g) * 50
    img, att = torch.randn(B * ncam, Ci, H, W, generator=g), torch.rand(B * ncam, H, W, generator=g)
    t = lambda x: x.to(dev).contiguous()                             # noqa: E731
    ind, mask, grid, feat, pinv, img, att = [t(x) for x in (ind, mask, grid, feat, pinv, img, att)]
    pos = torch.empty((ncam, n), dtype=torch.int32, device=dev)
    counts = torch.empty((B * ncam,), dtype=torch.int32, device=dev)
    P = lambda x: ctypes.c_void_p(x.data_ptr()) if x is not None else ctypes.c_void_p(0)   # noqa: E731
    _lib.check(lib.df3d_query_slots(P(mask), P(ind), n, B, ncam, P(pos), P(counts), ops._stream()))
    max_ne = int(counts.max())
    assert int(counts[2]) == 0

    def run(with_counts, use_att, use_pos):
        outs = [torch.full((B * ncam, max_ne, k), -3.0, device=dev) for k in (C, Ci, 2, 3, C)]
        if not with_counts:
            pass                                                    # the entry clears everything itself
        _lib.check(lib.df3d_assemble_queries2(P(feat), P(pinv), P(ind), P(grid), P(mask), P(pos), P(img), None, P(att) if use_att else None,
                                              n, C, Ci, B, ncam, H, W, max_ne, P(outs[0]), P(outs[1]), P(outs[2]), P(outs[3]),
                                              P(outs[4]) if use_pos else None, P(counts) if with_counts else None, ops._stream()))
        return outs[:4] + ([outs[4]] if use_pos else [])
    def run_slots(use_att, use_pos, ptr_table):
        """round 4: df3d_assemble_queries2_slots (a wave per slot, padding rows written by the same launch)"""
        outs = [torch.full((B * ncam, max_ne, k), -3.0, device=dev) for k in (C, Ci, 2, 3, C)]
        table = torch.empty((B * ncam * max_ne, 4), dtype=torch.int32, device=dev)
        ptrs = torch.tensor([img[i].data_ptr() for i in range(B * ncam)], dtype=torch.int64, device=dev) if ptr_table else None
        _lib.check(lib.df3d_assemble_queries2_slots(P(feat), P(pinv), P(ind), P(grid), P(mask), P(pos), None if ptr_table else P(img),
                                                    P(ptrs), P(att) if use_att else None, n, C, Ci, B, ncam, H, W, max_ne,
                                                    P(outs[0]), P(outs[1]), P(outs[2]), P(outs[3]), P(outs[4]) if use_pos else None,
                                                    P(counts), P(table), None, None, ops._stream()))
        return outs[:4] + ([outs[4]] if use_pos else [])
    for use_att, use_pos in ((True, True), (False, False)):
        a, b = run(True, use_att, use_pos), run(False, use_att, use_pos)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        for ptr_table in (False, True):
            for x, y in zip(run_slots(use_att, use_pos, ptr_table), b):
                assert torch.equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("C", [32, 128])
def test_gate_scatter_with_row_responses_inside(C):
    """df3d_gate_scatter_rows (round 3: the 9 tap responses of a voxel row computed inside the scatter, for the rows that win
    a pixel only) against torch.cat + GEMM + df3d_gate_scatter: same winners, S within fp32 summation noise; two scales
    accumulate into one S (clear only on the first)."""
    import ctypes
    from dualfusion import _lib
    lib = _lib.load()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(C)
    B, ncam, H, W, n = 2, 3, 24, 40, 5000
    p = lambda t: ctypes.c_void_p(t.data_ptr())       # noqa: E731
    outs = []
    for rows_inside in (False, True):
        S = torch.full((B * ncam, 9, H, W), 7.0, device=dev)
        gen.manual_seed(C)
        for scale in range(2):
            feats = torch.randn((n, C), generator=gen).to(dev)
            pinv = torch.randn((n, 3), generator=gen).to(dev)
            T = (torch.randn((9, C + 3), generator=gen) * 0.2).to(dev)
            ind = torch.zeros((n, 4), dtype=torch.int32)
            ind[:, 0] = torch.sort(torch.randint(0, B, (n,), generator=gen)).values.int()
            grid = torch.stack([torch.randint(-3, W + 3, (ncam, n), generator=gen),
                                torch.randint(-3, H + 3, (ncam, n), generator=gen)], -1).int().contiguous().to(dev)
            mask = (torch.rand((ncam, n), generator=gen) < 0.6).to(torch.uint8).to(dev)
            ind = ind.to(dev)
            winner = torch.empty((B * ncam, H, W), dtype=torch.int32, device=dev)
            if rows_inside:
                rc = lib.df3d_gate_scatter_rows(p(feats), C, p(pinv), p(T), p(ind), p(grid), p(mask), n, B, ncam, H, W,
                                                p(winner), p(S), int(scale == 0), None)
            else:
                s9 = (torch.cat([feats, pinv], 1) @ T.t()).contiguous()
                rc = lib.df3d_gate_scatter(p(s9), p(ind), p(grid), p(mask), n, B, ncam, H, W, p(winner), p(S),
                                           int(scale == 0), None)
            assert rc == 0
        torch.cuda.synchronize()
        outs.append((S.cpu(), winner.cpu()))
    assert torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][0].abs().max()) > 0.5
    assert float((outs[0][0] - outs[1][0]).abs().max()) < 1e-5 * max(1.0, float(outs[0][0].abs().max()))


@pytest.mark.parametrize("cin,cout", [(32, 32), (32, 64), (64, 64), (64, 128), (128, 128), (128, 256), (256, 256), (512, 64)])
def test_sparse_conv_three_part_precision(dev, cin, cout):
    """"split3" precision (round 4): operands in three bf16 parts (hi + mid + lo = the fp32 value exactly), six matrix-core
    products, fp32 accumulate -- against the float64 contraction: <= 4e-6 of the output scale (the exact-fp32 MFMA kernel
    sits at ~1e-6, the two-part split at ~1e-5), fused epilogue included; the emitted three-part rows equal
    split_rows(out) of the mode bit for bit and reconstruct the fp32 rows EXACTLY."""
    from dualfusion import ops
    shape, batch = [9, 48, 48], 2
    ind = detgen.clustered_voxels("c3s", batch, shape, n_seeds=8, walk=200)
    ind_t = T(ind, dev)
    feats = detgen.randn("c3f%d" % cin, (len(ind), cin))
    filt = detgen.randn("c3w%d_%d" % (cin, cout), (27, cin, cout), 0.5 / np.sqrt(cin))
    bias = detgen.randn("c3b%d" % cout, (cout,), 0.1)
    scale = 1 + detgen.randn("c3s%d" % cout, (cout,), 0.1)
    shift = detgen.randn("c3h%d" % cout, (cout,), 0.1)
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = "split3"
    try:
        assert ops.conv_split_supported(27, cin, cout) and ops.split_parts() == 3
        packed = ops.conv_pack_weights(T(filt, dev))
        fsplit = ops.split_rows(T(feats, dev))
        assert tuple(fsplit.shape) == (len(ind), 6 * cin)
        # hi + mid + lo == x exactly
        parts = fsplit.view(len(ind), cin // 8, 3, 8, 2).cpu().numpy().view(np.uint16).reshape(len(ind), cin // 8, 3, 8)
        recon = sum((parts[:, :, p].astype(np.uint32) << 16).view(np.float32).astype(np.float64) for p in range(3))
        assert np.array_equal(recon.reshape(len(ind), cin).astype(np.float32), feats)
        for subm in (1, 0):
            ks, st, pd, dl = [3, 3, 3], [2, 2, 2], [1, 1, 1], [1, 1, 1]
            outids, nbr, _ = _hip_rulebook(ind_t, batch, shape, ks, st, pd, dl, subm)
            n_out = outids.shape[0]
            res = detgen.randn("c3r%d" % cout, (n_out, cout))
            y, ys = ops.sparse_conv_split(fsplit, packed, nbr, n_out, cin, cout, bias=T(bias, dev), scale=T(scale, dev),
                                          shift=T(shift, dev), residual=T(res, dev), relu=True)
            nb = nbr.cpu().numpy()
            acc = np.zeros((n_out, cout), np.float64)
            for k in range(27):
                m = nb[k] >= 0
                acc[m] += feats[nb[k][m]].astype(np.float64) @ filt[k].astype(np.float64)
            ref = np.maximum((acc + bias) * scale + shift + res, 0)
            err = np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max()
            assert err < 4e-6, (subm, err)          # (fp32 accumulation over up to 27 x 512 terms; exact-fp32 MFMA kernel: < 5e-6)
            assert torch.equal(ys, ops.split_rows(y))
            # grouped / strided form (what the neck and the heads call): same result
            y2, _ = ops.conv_rows_split(fsplit, cin, 0, packed, cout, 1, nbr, n_out, T(bias, dev), T(scale, dev), T(shift, dev),
                                        relu=False)
            y1, _ = ops.sparse_conv_split(fsplit, packed, nbr, n_out, cin, cout, bias=T(bias, dev), scale=T(scale, dev),
                                          shift=T(shift, dev), relu=False, emit_split=False)
            assert torch.equal(y1, y2)
    finally:
        ops.CONV_PRECISION = old


def test_three_part_mode_end_to_end_detector():
    """The CenterPoint detector forward (backbone through the native executor, camera fusion, neck, head) in the "split3"
    mode against the exact-fp32 mode: every head map within 2e-5 of its scale (both are fp32-grade; the two-part split mode
    is ~1e-4), identical index sets."""
    from dualfusion import ops, synth
    from dualfusion.fusion import build_centerpoint_fusion, synthetic_camera_inputs
    from dualfusion.pipeline import CenterPointDetector
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = CenterPointDetector(fusion=build_centerpoint_fusion()).eval().to(dev)
    pts = [torch.from_numpy(synth.nusc_sweep(seed=9)).to(dev)]
    bd, ex = synthetic_camera_inputs(1, dev, seed=3, yaw_offset_deg=7.3)
    outs = {}
    old = ops.CONV_PRECISION
    try:
        for mode in ("fp32", "split3", "split"):
            ops.CONV_PRECISION = mode
            with torch.no_grad():
                x, multi = m.hot_path(pts, batch_dict=dict(bd), example=dict(ex))
                preds = m.bbox_head(x)
            outs[mode] = (multi["conv4"].indices.clone(), [{k: v.clone() for k, v in p.items()} for p in preds])
    finally:
        ops.CONV_PRECISION = old
    assert torch.equal(outs["fp32"][0], outs["split3"][0])
    worst3 = worst2 = 0.0
    for t in range(len(outs["fp32"][1])):
        for k, ref in outs["fp32"][1][t].items():
            s = float(ref.abs().max())
            worst3 = max(worst3, float((outs["split3"][1][t][k] - ref).abs().max()) / s)
            worst2 = max(worst2, float((outs["split"][1][t][k] - ref).abs().max()) / s)
    assert worst3 < 2e-5, (worst3, worst2)
    assert worst2 < 1e-3


def test_value_rows_on_a_side_stream_equal_the_in_line_chain(monkeypatch):
    """Round 4 (opt-in): the image side of ACTR (moments, GroupNorm fold, value rows) queued on its own stream beside the query
    assembly (ACTR.start_values, DF3D_VALUE_SIDE=1) -- bit-identical head maps to the single-stream order, over several frames
    (the buffers of both streams are recycled between them) and from a non-default caller stream."""
    from dualfusion import synth
    from dualfusion.fusion import build_centerpoint_fusion, synthetic_camera_inputs
    from dualfusion.pipeline import CenterPointDetector
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = CenterPointDetector(fusion=build_centerpoint_fusion()).eval().to(dev)
    frames = []
    for j in range(3):
        frames.append(([torch.from_numpy(synth.nusc_sweep(seed=20 + j)).to(dev)],
                       synthetic_camera_inputs(1, dev, seed=5 + j, yaw_offset_deg=3.0 * j)))

    def run(side):
        monkeypatch.setenv("DF3D_VALUE_SIDE", side)
        outs = []
        with torch.no_grad():
            for rnd in range(2):
                for pts, (bd, ex) in frames:
                    x, _ = m.hot_path(pts, batch_dict=dict(bd), example=dict(ex))
                    outs.append([v.clone() for p in m.bbox_head(x) for _, v in sorted(p.items())])
        torch.cuda.synchronize()
        return outs

    want = run("0")
    got = run("1")
    assert m.hot_path.fusion.pfat.__dict__.get("_value_stream") is not None      # the side stream was used
    for a, b in zip(want, got):
        assert len(a) == len(b) and all(torch.equal(x, y) for x, y in zip(a, b))
    other = torch.cuda.Stream(device=dev)
    other.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(other):
        got2 = run("1")
    other.synchronize()
    for a, b in zip(want, got2):
        assert all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.gpu
def test_pixel_major_query_rows_equal_the_channel_first_gather():
    """Round 4: df3d_query_pixel_rows (rank of every sampled pixel) + df3d_imgproj_split_compact (both projection kernels: the
    raw rows of the marked pixels, pixel-major, next to unchanged split rows / gate) + df3d_assemble_queries2_slots reading
    them: identical query tensors to the gather from the channel-first maps; every marked pixel's row equals its map column."""
    import ctypes
    from dualfusion import _lib, ops
    lib = _lib.load()
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(21)
    B, ncam, H, W, C, Ci, n = 1, 6, 30, 53, 128, 256, 6000
    S = H * W
    ind = torch.zeros(n, 4, dtype=torch.int32)
    mask = (torch.rand(ncam, n, generator=g) < 0.25).to(torch.uint8)
    grid = torch.stack([torch.randint(0, W, (ncam, n), generator=g), torch.randint(0, H, (ncam, n), generator=g)], 2).int()
    feat, pinv = torch.randn(n, C, generator=g), torch.rand(n, 3, generator=g) * 50
    img, att = torch.randn(B * ncam, Ci, H, W, generator=g), torch.rand(B * ncam, H, W, generator=g)
    t = lambda x: x.to(dev).contiguous()                             # noqa: E731
    ind, mask, grid, feat, pinv, img, att = [t(x) for x in (ind, mask, grid, feat, pinv, img, att)]
    P = lambda x: ctypes.c_void_p(x.data_ptr()) if x is not None else ctypes.c_void_p(0)   # noqa: E731
    pos = torch.empty((ncam, n), dtype=torch.int32, device=dev)
    counts = torch.empty((B * ncam,), dtype=torch.int32, device=dev)
    _lib.check(lib.df3d_query_slots(P(mask), P(ind), n, B, ncam, P(pos), P(counts), ops._stream()))
    max_ne = int(counts.max())
    nws = int(lib.df3d_query_pixel_rows_workspace_bytes(B, ncam, H, W))
    ws = torch.empty((nws,), dtype=torch.uint8, device=dev)
    pixrow = torch.empty((B * ncam, S), dtype=torch.int32, device=dev)
    total = torch.zeros((1,), dtype=torch.int32, device=dev)
    _lib.check(lib.df3d_query_pixel_rows(P(ind), P(grid), P(mask), n, B, ncam, H, W, P(pixrow), P(total), P(ws), nws, ops._stream()))
    total = int(total)
    marked = pixrow >= 0
    assert total == int(marked.sum()) and 0 < total < B * ncam * S
    assert sorted(pixrow[marked].tolist()) == list(range(total))
    ptrs = torch.tensor([img[i].data_ptr() for i in range(B * ncam)], dtype=torch.int64, device=dev)
    packed = ops.imgproj_pack(torch.randn(144, Ci, generator=g).to(dev) * 0.05)
    u0, g0 = ops.imgproj_split(ptrs, B * ncam, Ci, S, packed)
    rows_of = img.view(B * ncam, Ci, S).permute(0, 2, 1)[marked]                 # [total, Ci] in (image, pixel) order
    want_compact = torch.empty_like(rows_of)
    want_compact[pixrow[marked].long()] = rows_of
    for direct in ("0", "1"):
        os.environ["DF3D_IMGPROJ_DIRECT"] = direct
        try:
            u1, g1, compact = ops.imgproj_split(ptrs, B * ncam, Ci, S, packed, pixrow=pixrow, pixrow_total=total)
        finally:
            os.environ.pop("DF3D_IMGPROJ_DIRECT", None)
        assert torch.equal(u1, u0) and torch.equal(g1, g0) and torch.equal(compact, want_compact), direct

    def run(use_compact):
        outs = [torch.full((B * ncam, max_ne, k), -3.0, device=dev) for k in (C, Ci, 2, 3, C)]
        table = torch.empty((B * ncam * max_ne, 4), dtype=torch.int32, device=dev)
        _lib.check(lib.df3d_assemble_queries2_slots(P(feat), P(pinv), P(ind), P(grid), P(mask), P(pos), P(img), None, P(att), n, C, Ci,
                                                    B, ncam, H, W, max_ne, P(outs[0]), P(outs[1]), P(outs[2]), P(outs[3]), P(outs[4]),
                                                    P(counts), P(table), P(pixrow) if use_compact else None,
                                                    P(compact) if use_compact else None, ops._stream()))
        return outs
    for x, y in zip(run(True), run(False)):
        assert torch.equal(x, y)


@pytest.mark.gpu
def test_detector_with_pixel_major_query_rows_equals_plain(monkeypatch):
    """The CenterPoint + 3D-DF detector with DF3D_ASSEMBLE_COMPACT=1 (the frame head ranks the sampled pixels, the image
    projection writes their rows pixel-major, the by-slot assembly reads them) against the default gather: bit-identical head
    maps, frame after frame through the prefetching step()."""
    from dualfusion import synth
    from dualfusion.fusion import build_centerpoint_fusion, synthetic_camera_inputs
    from dualfusion.pipeline import CenterPointDetector
    dev = torch.device("cuda:0")

    def run(mode):
        monkeypatch.setenv("DF3D_ASSEMBLE_COMPACT", mode)
        frames = []
        for j in range(3):
            bd, ex = synthetic_camera_inputs(1, dev, seed=8 + j, yaw_offset_deg=2.0 * j)
            frames.append(([torch.from_numpy(synth.nusc_sweep(seed=30 + j)).to(dev)], bd, ex))
        torch.cuda.synchronize()
        torch.manual_seed(0)
        m = CenterPointDetector(fusion=build_centerpoint_fusion()).eval().to(dev)
        m.hot_path.resident_inputs = True                  # the clouds are complete in device memory (synchronised above)
        if m.hot_path.fusion is not None:
            m.hot_path.fusion.resident_inputs = True
        outs, used = [], 0
        with torch.no_grad():
            for rnd in range(2):
                for j, (pts, bd, ex) in enumerate(frames):
                    used += int(bool(m.hot_path.prefetch(pts, bd)))          # the frame head: pixel ranks come from it
                    x, _ = m.hot_path(pts, batch_dict=bd, example=ex)
                    outs.append([v.clone() for p in m.bbox_head(x) for _, v in sorted(p.items())])
        torch.cuda.synchronize()
        m.hot_path.close()
        return outs, used
    (want, _), (got, used) = run("0"), run("1")
    assert used >= 4, used
    for a, b in zip(want, got):
        assert len(a) == len(b) and all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.gpu
@pytest.mark.parametrize("cin,cout,strided", [(5, 16, False), (16, 16, False), (16, 32, True)])
def test_small_channel_conv_from_row_lists(cin, cout, strided):
    """Round 4: df3d_nbr_row_lists (the present entries of every output row of a neighbour table: offsets, packed entries)
    against numpy, and df3d_sparse_conv_fused_lists against df3d_sparse_conv_fused on the same table: bit-identical rows
    (bias, folded BN, residual, ReLU in the epilogue), submanifold and stride-2 geometry of a nuScenes-shaped sweep."""
    import ctypes
    from dualfusion import _lib, ops, synth
    lib = _lib.load()
    dev = torch.device("cuda:0")
    P = lambda x: ctypes.c_void_p(x.data_ptr()) if x is not None else ctypes.c_void_p(0)   # noqa: E731
    pts = synth.nusc_sweep(seed=3)
    _, c, _, _ = ops.hard_voxelize(torch.from_numpy(pts).to(dev), synth.NUSC_VOXEL, synth.NUSC_RANGE, 10, 60000)
    shape = [41, 1440, 1440]
    ind = torch.cat([torch.zeros((c.shape[0], 1), dtype=torch.int32, device=dev), c], 1).contiguous()
    n_in = ind.shape[0]
    if strided:
        oshape = [(v + 2 - 3) // 2 + 1 for v in shape]
        out_ind, _ = ops.conv_out_indices(ind, 1, shape, oshape, [3, 3, 3], [2, 2, 2], [1, 1, 1])
        out_ind = out_ind.contiguous()
        grid = ops.grid_build(ind, 1, shape)
        nbr = ops.conv_neighbors(grid, out_ind, [3, 3, 3], [2, 2, 2], [1, 1, 1])
        n_out = out_ind.shape[0]
    else:
        grid = ops.grid_build(ind, 1, shape)
        nbr = ops.subm_neighbors(grid, ind, [3, 3, 3])
        n_out = n_in
    K = nbr.shape[0]
    nb = int(lib.df3d_nbr_row_lists_bytes(K, n_out))
    blob = torch.zeros((nb,), dtype=torch.uint8, device=dev)
    _lib.check(lib.df3d_nbr_row_lists(P(nbr), K, n_out, n_in, P(blob), nb, ops._stream()))
    torch.cuda.synchronize()
    t = nbr.cpu().numpy()                                               # [K, n_out]
    off = blob[:(n_out + 1) * 4].view(torch.int32).cpu().numpy().astype(np.int64)
    want_cnt = (t >= 0).sum(0)
    assert off[0] == 0 and np.array_equal(np.diff(off), want_cnt)
    off_bytes = ((n_out + 1) * 4 + 255) // 256 * 256
    ent = blob[off_bytes:off_bytes + int(off[-1]) * 4].view(torch.int32).cpu().numpy().astype(np.int64)
    ks, rows = np.nonzero(t.T >= 0)[1], np.nonzero(t.T >= 0)[0]           # row-major walk: rows ascending, offsets ascending
    assert np.array_equal(ent >> 26, ks) and np.array_equal(ent & 0x3ffffff, t.T[rows, ks])
    g = torch.Generator().manual_seed(cin * 100 + cout)
    feat = torch.randn(n_in, cin, generator=g).to(dev)
    w = (torch.randn(K, cin, cout, generator=g) * 0.2).to(dev)
    bias, scale, shift = [torch.randn(cout, generator=g).to(dev) for _ in range(3)]
    res = torch.randn(n_out, cout, generator=g).to(dev)
    for use_res, relu in ((True, 1), (False, 0)):
        want = torch.empty((n_out, cout), device=dev)
        got = torch.full((n_out, cout), -7.0, device=dev)
        _lib.check(lib.df3d_sparse_conv_fused(P(feat), n_in, cin, P(w), K, cout, P(nbr), n_out, P(bias), P(scale), P(shift),
                                              P(res) if use_res else None, relu, P(want), ops._stream()))
        gsplit = torch.zeros((n_out, 4 * cout), dtype=torch.uint8, device=dev)
        _lib.check(lib.df3d_sparse_conv_fused_lists(P(feat), n_in, cin, P(w), K, cout, P(nbr), P(blob), n_out, P(bias), P(scale),
                                                    P(shift), P(res) if use_res else None, relu, P(got), P(gsplit),
                                                    ops._stream()))
        assert torch.equal(got, want), (use_res, relu)
        assert torch.equal(gsplit, ops.split_rows(want))          # the operand split of the rows from the same launch
        # without lists the entry is df3d_sparse_conv_fused + df3d_split_rows
        got2, gsplit2 = torch.empty_like(got), torch.zeros_like(gsplit)
        _lib.check(lib.df3d_sparse_conv_fused_lists(P(feat), n_in, cin, P(w), K, cout, P(nbr), None, n_out, P(bias), P(scale),
                                                    P(shift), P(res) if use_res else None, relu, P(got2), P(gsplit2),
                                                    ops._stream()))
        assert torch.equal(got2, want) and torch.equal(gsplit2, gsplit)
    assert float(want.abs().max()) > 0.1


@pytest.mark.gpu
def test_relu_dropout_in_place_forward_and_maskless_backward():
    """df3d_relu_dropout: h <- relu(h) . keep / (1 - p) in place (the hidden rows of forward_ffn, actr_transformer.py:309-311 /
    388-395, in a training step).  p = 0 is torch's relu bit for bit, forward and backward; p = 0.1 keeps 90 % of the active
    elements (binomial bounds), scales them by 1 / 0.9, repeats under the same seed and differs under another; the backward passes
    grad / (1 - p) exactly where the forward kept a positive value -- checked against the mask read off the forward's result."""
    from dualfusion import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(3)
    x = torch.randn((3001, 1024), generator=gen)
    x[5, :7] = 0.0
    g = torch.randn(x.shape, generator=gen).to(dev)
    a = x.to(dev).requires_grad_(True)
    y = ops.relu_dropout_(a * 1.0, 0.0)
    y.backward(g)
    b = x.to(dev).requires_grad_(True)
    yr = torch.relu(b)
    yr.backward(g)
    assert torch.equal(y, yr) and torch.equal(a.grad, b.grad)
    p = 0.1
    outs = []
    for seed in (11, 11, 12):
        h = x.to(dev).requires_grad_(True)
        y = ops.relu_dropout_(h * 1.0, p, seed=seed)
        y.backward(g)
        outs.append((y.detach().cpu(), h.grad.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert not torch.equal(outs[0][0], outs[2][0])
    y, gx = outs[0]
    active = x > 0
    kept = y != 0
    assert not bool((kept & ~active).any())
    n, k = int(active.sum()), int(kept.sum())
    assert abs(k - (1 - p) * n) <= 5 * (n * p * (1 - p)) ** 0.5, (k, n)                 # five sigma of the binomial
    scale = np.float32(1.0 / (1.0 - np.floor(p * 16777216.0) / 16777216.0))
    assert torch.equal(y[kept], x[kept] * float(scale))
    assert torch.equal(gx, torch.where(kept, g.cpu() * float(scale), torch.zeros(())))
    # the mask does not depend on the launch shape: the first 1001 elements of a longer call = a call on 1001 elements
    h1, h2 = x.reshape(-1)[:1001].clone().to(dev), x.reshape(-1)[:4004].clone().to(dev)
    y1, y2 = ops.relu_dropout_(h1, 0.3, seed=5), ops.relu_dropout_(h2, 0.3, seed=5)
    assert torch.equal(y1, y2[:1001])
    # without an explicit seed the mask follows torch's device generator: repeats under manual_seed, moves from call to call
    runs = []
    for _ in range(2):
        torch.manual_seed(123)
        runs.append([ops.relu_dropout_(x.to(dev), 0.1).cpu() for _ in range(2)])
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1]) and not torch.equal(runs[0][0], runs[0][1])
    # per-row independence: no column of a [rows, 1024] tensor is dropped much more often than p
    col = (outs[0][0] != 0).float().sum(0) / active.float().sum(0).clamp_min(1)
    assert float(col.min()) > 0.8 and float(col.max()) <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("rows,C", [(5003, 128), (777, 64), (300, 256), (1030, 32)])
def test_dropout_add_layernorm_forward_and_backward(rows, C):
    """df3d_dropout_add_layernorm: norm(x + dropout(y)) of the encoder layers' residual steps (actr_transformer.py:311-312,
    389-396, 416-417) as one kernel each way.  p = 0 against nn.LayerNorm in float64 (output, d x = d y, d gamma, d beta);
    p = 0.2 against the torch composition with the SAME mask (the hash of df3d_relu_dropout over the element index, read off a
    relu_dropout_ call with that seed)."""
    from dualfusion import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(rows + C)
    x, y = torch.randn((3, rows, C), generator=gen), torch.randn((3, rows, C), generator=gen) * 2 + 0.3
    g = torch.randn((3, rows, C), generator=gen)
    norm = torch.nn.LayerNorm(C)
    with torch.no_grad():
        norm.weight.copy_(torch.rand(C, generator=gen) + 0.5)
        norm.bias.copy_(torch.randn(C, generator=gen))
    for p in (0.0, 0.2):
        drop = torch.nn.Dropout(p)
        nd = copy.deepcopy(norm).to(dev)
        xa, ya = x.to(dev).requires_grad_(True), y.to(dev).requires_grad_(True)
        torch.manual_seed(99)
        seed = ops._dropout_seed(dev) if p else 0
        torch.manual_seed(99)
        out = ops.dropout_add_layernorm(xa, ya, nd, drop)
        assert type(out.grad_fn).__name__ == "_DropoutAddLayerNormBackward"
        out.backward(g.to(dev))
        mask = torch.ones_like(x, dtype=torch.float64)
        if p:
            kept = ops.relu_dropout_(torch.ones(x.numel(), device=dev), p, seed=seed).cpu().view(x.shape)
            assert 0.7 < float((kept != 0).double().mean()) < 0.9
            mask = kept.double()                                    # 0 or 1 / (1 - p)
        n64 = copy.deepcopy(norm).double()
        xr, yr = x.double().requires_grad_(True), y.double().requires_grad_(True)
        ref = n64(xr + yr * mask)
        ref.backward(g.double())
        for a, b, name in ((out, ref, "out"), (xa.grad, xr.grad, "dx"), (ya.grad, yr.grad, "dy"),
                           (nd.weight.grad, n64.weight.grad, "dgamma"), (nd.bias.grad, n64.bias.grad, "dbeta")):
            err = float((a.detach().cpu().double() - b.detach()).abs().max()) / max(1.0, float(b.detach().abs().max()))
            assert err <= (2e-5 if name in ("dgamma", "dbeta") else 5e-6), (name, p, err)


@pytest.mark.gpu
def test_bf16_feed_forward_branch_pieces():
    """The bf16 mixed-precision feed-forward branch of a training step (BASELINE configs[2] / [3]): `linear_bf16` (bfloat16
    operands, fp32 accumulation, the weight gradient as per-sample batched products) against float64 within bfloat16's rounding,
    and df3d_relu_dropout_bf16 (in place on bfloat16 rows) with the same mask as the fp32 kernel under the same seed."""
    from dualfusion import ops
    from dualfusion.linear_rows import linear_bf16
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(17)
    for shape in ((6, 3001, 128), (32768, 128)):
        x = torch.randn(shape, generator=gen)
        w, b = torch.randn((256, 128), generator=gen) * 0.1, torch.randn(256, generator=gen)
        g = torch.randn(shape[:-1] + (256,), generator=gen)
        xa, wa, ba = (t.to(dev).requires_grad_(True) for t in (x, w, b))
        y = linear_bf16(xa, wa, ba)
        assert y.dtype == torch.bfloat16
        y.backward(g.to(dev).to(torch.bfloat16))
        xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
        yr = torch.nn.functional.linear(xr, wr, br)
        yr.backward(g.double())
        for a, r, name in ((y, yr, "y"), (xa.grad, xr.grad, "dx"), (wa.grad, wr.grad, "dw"), (ba.grad, br.grad, "db")):
            assert a.dtype == (torch.bfloat16 if name == "y" else torch.float32)
            err = float((a.detach().cpu().double() - r.detach()).abs().max()) / float(r.detach().abs().max())
            assert err <= 2e-2, (name, shape, err)                                     # 8 significant bits per operand
    h = torch.randn((4099, 1024), generator=gen)
    h16 = h.to(dev).to(torch.bfloat16)
    want = ops.relu_dropout_(h16.float(), 0.1, seed=77)                                # fp32 kernel on the rounded values
    got = ops.relu_dropout_((h16.clone().requires_grad_(True)) * 1, 0.1, seed=77)
    assert got.dtype == torch.bfloat16 and torch.equal(got.detach().float() != 0, want != 0)
    assert float((got.detach().float() - want).abs().max()) <= 2 ** -7 * float(want.abs().max())
    gg = torch.randn(h.shape, generator=gen).to(dev).to(torch.bfloat16)
    leaf = h16.clone().requires_grad_(True)
    out = ops.relu_dropout_(leaf * 1, 0.1, seed=77)
    out.backward(gg)
    scale = 1.0 / (1.0 - np.floor(0.1 * 16777216.0) / 16777216.0)
    ref = torch.where(out.detach() != 0, gg.float() * scale, torch.zeros((), device=dev))
    assert leaf.grad.dtype == torch.bfloat16 and float((leaf.grad.float() - ref).abs().max()) <= 2 ** -7 * float(ref.abs().max())
