"""CenterPoint sparse backbone (SpMiddleResNetFHD) on the MI355X modules vs the oracle composition,
same state_dict, same synthetic sweep.  fp32, tolerance 1e-3 (north_star)."""
import numpy as np
import pytest

import detgen
import oracle_models as om
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _model_and_sd(dev):
    from dualfusion.backbones import SpMiddleResNetFHD
    model = SpMiddleResNetFHD(num_input_features=5).eval()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = detgen.det_state_dict(shapes)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return model.to(dev), sd


def test_parameter_names_match_reference_checkpoint_layout():
    from dualfusion.backbones import SpMiddleResNetFHD
    sd = SpMiddleResNetFHD(num_input_features=5).state_dict()
    # SURVEY.md Appendix B
    assert tuple(sd["conv_input.0.weight"].shape) == (3, 3, 3, 5, 16)
    assert tuple(sd["conv2.0.weight"].shape) == (3, 3, 3, 16, 32)
    assert tuple(sd["conv4.0.weight"].shape) == (3, 3, 3, 64, 128)
    assert tuple(sd["extra_conv.0.weight"].shape) == (3, 1, 1, 128, 128)
    assert "conv1.0.conv1.bias" in sd and "conv_input.0.bias" not in sd
    assert "conv3.4.bn2.running_var" in sd


@pytest.mark.parametrize("batch", [1, 2])
def test_centerpoint_backbone_vs_oracle(batch):
    from dualfusion import ops, synth
    dev = torch.device("cuda:0")
    model, sd = _model_and_sd(dev)
    # a reduced grid keeps the oracle's dense rulebook grids small: 16 m x 16 m around the sensor
    rng = [-9.6, -9.6, -5.0, 9.6, 9.6, 3.0]
    feats, coors = [], []
    o_feats, o_coors = [], []
    for b in range(batch):
        pts = synth.nusc_sweep(seed=20 + b)
        v, c, n, mean = ops.hard_voxelize(torch.from_numpy(pts).to(dev), synth.NUSC_VOXEL, rng, 10, 120000)
        ov, oc, on = orc.hard_voxelize(pts, synth.NUSC_VOXEL, rng, 10, 120000)
        assert np.array_equal(c.cpu().numpy(), oc)
        feats.append(mean)
        coors.append(torch.cat([torch.full((c.shape[0], 1), b, dtype=torch.int32, device=dev), c], 1))
        o_feats.append(orc.mean_vfe(ov, on))
        o_coors.append(np.concatenate([np.full((len(oc), 1), b, np.int32), oc], 1))
    grid_xyz = [256, 256, 40]
    with torch.no_grad():
        dense, ms = model(torch.cat(feats), torch.cat(coors), batch, grid_xyz)
    o_dense, o_ms = om.centerpoint_backbone(sd, np.concatenate(o_feats), np.concatenate(o_coors), batch, grid_xyz)
    assert tuple(dense.shape) == o_dense.shape == (batch, 256, 32, 32)
    for name in ("conv1", "conv2", "conv3", "conv4"):
        mi, mf = om.sort_rows(ms[name].indices.cpu().numpy(), ms[name].features.cpu().numpy())
        oi, of = om.sort_rows(o_ms[name].indices, o_ms[name].features)
        assert np.array_equal(mi, oi), name                   # voxel sets bit-exact
        scale = np.abs(of).max()
        assert np.abs(mf - of).max() <= 1e-3 * max(scale, 1.0), (name, np.abs(mf - of).max(), scale)
    np.testing.assert_allclose(dense.cpu().numpy(), o_dense, rtol=1e-3, atol=1e-3 * max(np.abs(o_dense).max(), 1.0))
