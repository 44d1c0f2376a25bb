"""Evaluation metrics: oracle pinned to the reference classes' own measures (CPU); device metrics vs both (GPU)."""
import os

import numpy as np
import pytest
import torch

import metrics_oracle as mo
from util import GOLDEN


def _inputs(seed=7, B=6):
    # same generator as tests/golden/make_golden.py::metric_inputs
    g = torch.Generator().manual_seed(seed)
    gt = 0.08 * torch.randn(B, 799, 3, generator=g) + torch.tensor([0.0, 0.0, 0.6])
    ang = 0.15 * torch.randn(B, generator=g)
    R = torch.stack([torch.tensor([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
                     for a in ang.tolist()])
    c = gt.mean(1, keepdim=True)
    pred = ((gt - c) @ R.transpose(1, 2)) * (1.0 + 0.05 * torch.randn(B, 1, 1, generator=g)) + c
    pred = pred + 0.004 * torch.randn(B, 799, 3, generator=g) + 0.01 * torch.randn(B, 1, 3, generator=g)
    return pred, gt


def _golden():
    return np.load(os.path.join(GOLDEN, "metrics.npz"))


def test_oracle_pa_matches_reference_measures():
    z = _golden()
    pred, gt = _inputs()
    p, g = pred.numpy(), gt.numpy()
    m = mo.pa_measures(p[:, :21], g[:, :21], p[:, 21:], g[:, 21:])
    for k in ("pa_mpjpe", "mpjpe", "pa_mpvpe", "mpvpe"):
        assert abs(m[k] - float(z["pa." + k])) < 2e-8, k


def test_oracle_pck_matches_reference_measures():
    z = _golden()
    pred, gt = _inputs()
    p, g = pred.numpy(), gt.numpy()
    for tag, sl in (("j", slice(0, 21)), ("v", slice(21, 799))):
        m = mo.pck_measures(p[:, sl], g[:, sl], 0.0, 0.05, 20)
        assert np.array_equal(m["pck_curve_per_kp"], z[f"{tag}.pck_curve_per_kp"])
        assert np.max(np.abs(m["auc_per_kp"] - z[f"{tag}.auc_per_kp"])) < 1e-12
        assert np.max(np.abs(m["epe_mean_per_kp"] - z[f"{tag}.epe_mean_per_kp"])) < 1e-8
        assert abs(m["pck_002"] - float(z[f"{tag}.pck_002"])) < 1e-12


@pytest.mark.gpu
def test_device_pa_eval_matches_reference():
    from poem_v2_amd.metrics import PAEval
    z = _golden()
    pred, gt = _inputs()
    pa = PAEval(None, mesh_score=True, device="cuda:0")
    for sl in (slice(0, 4), slice(4, 6)):                     # two feeds, as in the golden run
        pa.feed(pred[sl, :21].cuda(), gt[sl, :21].cuda(), pred[sl, 21:].cuda(), gt[sl, 21:].cuda())
    m = pa.reduce().get_measures()
    for k in ("pa_mpjpe", "mpjpe", "pa_mpvpe", "mpvpe"):
        assert abs(m[k] - float(z["pa." + k])) < 2e-7, (k, m[k], float(z["pa." + k]))     # metres (2e-4 mm)
    assert "pa_mpjpe(mm)" in str(pa) and abs(pa.get_result() - m["pa_mpjpe"]) == 0


@pytest.mark.gpu
def test_device_pa_is_invariant_to_similarity_transforms():
    """Size-independent property at full batch size: PA error of a rotated / scaled / shifted copy of the GT is ~0 and
    PA error is unchanged when the prediction is moved by a similarity transform."""
    from poem_v2_amd.metrics import PAEval
    g = torch.Generator().manual_seed(1)
    gt = 0.08 * torch.randn(32, 778, 3, generator=g)
    q, _ = torch.linalg.qr(torch.randn(32, 3, 3, generator=g))
    moved = (gt @ q) * 1.3 + torch.tensor([0.1, -0.2, 0.3])
    pa = PAEval(None, device="cuda:0")
    pa.feed(moved.cuda(), gt.cuda())
    assert pa.get_measures()["pa_mpjpe"] < 1e-7
    noisy = gt + 0.01 * torch.randn(gt.shape, generator=g)
    a, b = PAEval(None, device="cuda:0"), PAEval(None, device="cuda:0")
    a.feed(noisy.cuda(), gt.cuda())
    b.feed(((noisy @ q) * 0.7 + 0.05).cuda(), gt.cuda())
    assert abs(a.get_measures()["pa_mpjpe"] - b.get_measures()["pa_mpjpe"]) < 1e-7


@pytest.mark.gpu
def test_device_pck_matches_reference():
    from poem_v2_amd.metrics import Joint3DPCK, Vert3DPCK
    z = _golden()
    pred, gt = _inputs()
    cfg = dict(VAL_MIN=0.0, VAL_MAX=0.05, STEPS=20)
    jp, vp = Joint3DPCK(device="cuda:0", EVAL_TYPE="joints_3d", **cfg), Vert3DPCK(device="cuda:0", EVAL_TYPE="verts_3d", **cfg)
    for sl in (slice(0, 4), slice(4, 6)):
        jp.feed({"pred_joints_3d": pred[sl, :21].cuda()}, {"master_joints_3d": gt[sl, :21].cuda()})
        vp.feed({"pred_verts_3d": pred[sl, 21:].cuda()}, {"master_verts_3d": gt[sl, 21:].cuda()})
    for tag, m in (("j", jp), ("v", vp)):
        ms = m.reduce().get_measures()
        assert np.array_equal(ms["pck_curve_per_kp"], z[f"{tag}.pck_curve_per_kp"])        # integer counts: exact
        assert np.max(np.abs(ms["auc_per_kp"] - z[f"{tag}.auc_per_kp"])) < 1e-12
        assert np.max(np.abs(ms["epe_mean_per_kp"] - z[f"{tag}.epe_mean_per_kp"])) < 1e-8
        assert abs(ms["auc_all"] - float(z[f"{tag}.auc_all"])) < 1e-12
        assert abs(m.get_pck_all(0.02) - float(z[f"{tag}.pck_002"])) < 1e-12
        assert np.array_equal(ms["thresholds"], z[f"{tag}.thresholds"])
    with pytest.raises(ValueError):
        Joint3DPCK(device="cuda:0", EVAL_TYPE="nope", **cfg)
    with pytest.raises(RuntimeError):
        Joint3DPCK(device="cpu", EVAL_TYPE="joints_3d", **cfg).feed({"pred_joints_3d": pred[:, :21]}, {"master_joints_3d": gt[:, :21]})


# ---- mano_to_openpose (joints from the mesh, lib/utils/transform.py:836-872) ----------------------------------------
def _openpose_golden():
    return np.load(os.path.join(GOLDEN, "openpose.npz"))["joints"]


def test_oracle_mano_to_openpose_matches_reference():
    ref = _openpose_golden()
    out = mo.mano_to_openpose(mo.synthetic_j_regressor(11), mo.synthetic_mano_verts(3, 11))
    assert out.shape == (3, 21, 3)
    assert np.max(np.abs(out - ref)) < 2e-7
    # tips are exact copies of vertices at the OpenPose tip slots 4, 8, 12, 16, 20
    V = mo.synthetic_mano_verts(3, 11)
    assert np.array_equal(out[:, [4, 8, 12, 16, 20]], V[:, list(mo.MANO_TIP_VERTICES)])


@pytest.mark.gpu
def test_device_mano_to_openpose_matches_reference():
    from poem_v2_amd.metrics import mano_to_openpose
    ref = _openpose_golden()
    J, V = mo.synthetic_j_regressor(11), mo.synthetic_mano_verts(3, 11)
    out = mano_to_openpose(torch.from_numpy(J).cuda(), torch.from_numpy(V).cuda()).cpu().numpy()
    assert np.max(np.abs(out - ref)) < 2e-7                                  # metres
    assert np.max(np.abs(out - mo.mano_to_openpose(J, V))) < 2e-7
    assert np.array_equal(out[:, [4, 8, 12, 16, 20]], V[:, list(mo.MANO_TIP_VERTICES)])
    # a sample's joints do not depend on the batch it sits in (bit-exact), full-size batch
    Vb = mo.synthetic_mano_verts(256, 5)
    big = mano_to_openpose(torch.from_numpy(J).cuda(), torch.from_numpy(Vb).cuda())
    one = mano_to_openpose(torch.from_numpy(J).cuda(), torch.from_numpy(Vb[100:101]).cuda())
    assert torch.equal(big[100:101], one)
    with pytest.raises(RuntimeError):
        mano_to_openpose(torch.from_numpy(J), torch.from_numpy(V))           # CPU tensors: no fallback
