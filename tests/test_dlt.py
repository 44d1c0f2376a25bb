"""DLT triangulation: oracle pinned to the reference's own outputs (CPU), HIP kernel vs both (GPU)."""
import importlib.util
import json
import os

import numpy as np
import pytest
import torch

import dlt_oracle as do
from util import GOLDEN

_spec = importlib.util.spec_from_file_location("make_golden_inputs", os.path.join(GOLDEN, "make_golden.py"))


def _dlt_inputs(views, seed):
    # same generator as the golden script (kept in one place: tests/golden/make_golden.py::dlt_inputs)
    import poem_v2_amd as pk
    b = pk.inputs.synthetic_batch(views, seed=seed)
    m = b["img_metas"]
    K, E = m["cam_intr"], m["cam_extr"]
    T = torch.linalg.inv(E)
    vs = torch.repeat_interleave(torch.arange(len(views)), torch.tensor(views))
    X = b["reference_joints"][vs]
    pc = (T[:, None, :3, :3] @ X[..., None]).squeeze(-1) + T[:, None, :3, 3]
    q = (K[:, None] @ pc[..., None]).squeeze(-1)
    uv = q[..., :2] / q[..., 2:]
    g = torch.Generator().manual_seed(seed + 99)
    return uv + 1.5 * torch.randn(uv.shape, generator=g), K, E


def _golden():
    z = np.load(os.path.join(GOLDEN, "dlt.npz"))
    return z, json.loads(bytes(z["meta"]).decode())["cases"]


@pytest.mark.parametrize("name", ["u8", "u2", "ragged"])
def test_oracle_matches_reference_outputs(name):
    z, cases = _golden()
    views, seed = cases[name]["views"], cases[name]["seed"]
    uv, K, E = _dlt_inputs(views, seed)
    got = do.triangulate_reference_joints(uv, K, E, views)
    assert float((got - torch.from_numpy(z[name])).abs().max()) < 2e-6            # metres (fp32 SVD both sides)
    if name + ".batched" in z:
        B, N = len(views), views[0]
        T = torch.linalg.inv(E)
        gb = do.batch_triangulate_dlt(uv.view(B, N, 21, 2), K.view(B, N, 3, 3), T.view(B, N, 4, 4))
        assert float((gb - torch.from_numpy(z[name + ".batched"])).abs().max()) < 2e-6


def test_oracle_recovers_noise_free_points():
    views = [4, 7]
    import poem_v2_amd as pk
    b = pk.inputs.synthetic_batch(views, seed=3)
    uv, K, E = _dlt_inputs(views, 3)
    # noise-free projections triangulate back to the joints they came from
    T = torch.linalg.inv(E)
    vs = torch.repeat_interleave(torch.arange(2), torch.tensor(views))
    X = b["reference_joints"][vs]
    pc = (T[:, None, :3, :3] @ X[..., None]).squeeze(-1) + T[:, None, :3, 3]
    q = (K[:, None] @ pc[..., None]).squeeze(-1)
    clean = q[..., :2] / q[..., 2:]
    got = do.triangulate_reference_joints(clean, K, E, views)
    assert float((got - b["reference_joints"]).abs().max()) < 5e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["u8", "u2", "ragged"])
def test_hip_dlt_matches_reference_and_oracle(name):
    import poem_v2_amd as pk
    z, cases = _golden()
    views, seed = cases[name]["views"], cases[name]["seed"]
    uv, K, E = _dlt_inputs(views, seed)
    got = pk.triangulation.triangulate_reference_joints(uv.cuda(), K.cuda(), E.cuda(), views).cpu()
    ref = torch.from_numpy(z[name])
    # tolerance: 5e-6 m = 5e-3 mm (the reference's own fp32 SVD sits ~5e-7 m from the fp64 solution)
    assert float((got - ref).abs().max()) < 5e-6
    assert float((got - do.triangulate_reference_joints(uv, K, E, views)).abs().max()) < 5e-6
    if name + ".batched" in z:
        B, N = len(views), views[0]
        T = torch.linalg.inv(E)
        gb = pk.triangulation.batch_triangulate_dlt_torch(uv.view(B, N, 21, 2).cuda(), K.view(B, N, 3, 3).cuda(),
                                                          T.view(B, N, 4, 4).cuda()).cpu()
        assert float((gb - torch.from_numpy(z[name + ".batched"])).abs().max()) < 5e-6


@pytest.mark.gpu
def test_hip_dlt_full_batch_properties():
    """BASELINE-size batch (32 samples, ragged 2..10 views): per-sample independence and noise-free round trip."""
    import poem_v2_amd as pk
    rng = np.random.RandomState(5)
    views = rng.randint(2, 11, size=32).tolist()
    uv, K, E = _dlt_inputs(views, 77)
    full = pk.triangulation.triangulate_reference_joints(uv.cuda(), K.cuda(), E.cuda(), views)
    offs = np.concatenate([[0], np.cumsum(views)])
    for i in (0, 13, 31):
        s, e = offs[i], offs[i + 1]
        one = pk.triangulation.triangulate_reference_joints(uv[s:e].cuda(), K[s:e].cuda(), E[s:e].cuda(), [views[i]])
        assert torch.equal(one[0], full[i])
    orc = do.triangulate_reference_joints(uv, K, E, views)
    assert float((full.cpu() - orc).abs().max()) < 5e-6
    with pytest.raises(RuntimeError):
        pk.triangulation.triangulate_reference_joints(uv, K, E, views)       # CPU tensors: no fallback
    with pytest.raises(ValueError):
        pk.triangulation.triangulate_reference_joints(uv[:1].cuda(), K[:1].cuda(), E[:1].cuda(), [1])


def _heatmaps():
    g = torch.Generator().manual_seed(44)
    return torch.sigmoid(4.0 * torch.randn(5, 21, 32, 32, generator=g) - 3.0)


def test_oracle_heatmap_readout_matches_reference():
    z, _ = _golden()
    got = do.heatmap_to_uv(_heatmaps(), 256.0, 256.0)
    assert float((got - torch.from_numpy(z["hm_uv"])).abs().max()) < 2e-4      # pixels


@pytest.mark.gpu
def test_hip_heatmap_readout_and_chain():
    import poem_v2_amd as pk
    z, _ = _golden()
    hm = _heatmaps()
    got = pk.triangulation.heatmap_to_uv(hm.cuda(), 256.0, 256.0).cpu()
    assert float((got - torch.from_numpy(z["hm_uv"])).abs().max()) < 2e-4      # pixels
    # a one-hot map reads out its own pixel (x / W * img_w)
    one = torch.zeros(1, 2, 32, 32)
    one[0, 0, 5, 9] = 1.0
    one[0, 1, 31, 0] = 0.5
    uv = pk.triangulation.heatmap_to_uv(one.cuda(), 256.0, 128.0).cpu()
    assert torch.allclose(uv[0, 0], torch.tensor([9 / 32 * 256.0, 5 / 32 * 128.0]), atol=1e-3)
    assert torch.allclose(uv[0, 1], torch.tensor([0.0, 31 / 32 * 128.0]), atol=1e-3)
    # chained: heat maps -> uv -> DLT equals the two-step oracle
    views = [2, 3]
    b = pk.inputs.synthetic_batch(views, seed=9)
    m = b["img_metas"]
    rj = pk.triangulation.reference_joints_from_heatmaps(hm.cuda(), m["cam_intr"].cuda(), m["cam_extr"].cuda(), views, 256, 256)
    orc = do.triangulate_reference_joints(do.heatmap_to_uv(hm, 256.0, 256.0), m["cam_intr"], m["cam_extr"], views)
    assert rj.shape == (2, 21, 3)
    err = (rj.cpu() - orc).abs()
    assert float(err.max()) < 5e-3 * max(1.0, float(orc.abs().max()))       # random maps: ill-conditioned rays, loose bound
