"""SURVEY a19, the MANO half of the parametric tail: rot6d -> axis-angle and the MANO layer.

Both third-party (pytorch3d.transforms, manotorch) and absent -> parity unpinned; pinned here to MATHEMATICS instead of to
the oracle's restatement: Rodrigues(axis-angle) must reproduce the Gram-Schmidt rotation the six numbers define (incl.
angles within 1e-3 of 0 and of pi and every ``q_abs`` branch of the quaternion extraction); the MANO layer must satisfy the
model's defining properties.  CPU: the oracle's functions.  GPU (``-m gpu``): the HIP kernels through the C ABI, against
the same mathematics and against the independently formulated oracle."""
import math

import numpy as np
import pytest
import torch

import mano_oracle as mo
import poem_oracle as po

DEV = "cuda:0"


def _rot6d_cases():
    """(n,6) six-dimensional rotations with the rotation they encode chosen by axis / angle: generic, angle ~ 0, angle ~ pi
    (both sides), and axes that make each of the four quaternion candidates the largest."""
    g = torch.Generator().manual_seed(0)
    axes = torch.nn.functional.normalize(torch.randn(40, 3, generator=g, dtype=torch.float64), dim=-1)
    axes = torch.cat([axes, torch.eye(3, dtype=torch.float64), -torch.eye(3, dtype=torch.float64)])
    angles = [0.0, 1e-7, 1e-4, 9e-4, 0.3, 1.0, math.pi / 2, 2.0, 3.0, math.pi - 9e-4, math.pi - 1e-4, math.pi - 1e-6]
    aa = torch.stack([a * t for a in axes for t in angles])
    R = mo.rodrigues(aa)                                          # fp64 ground-truth rotations
    # six numbers = first two ROWS of R (pytorch3d convention), scaled / sheared so that Gram-Schmidt has work to do
    s1 = 0.5 + torch.rand(len(R), 1, generator=g, dtype=torch.float64) * 2
    s2 = 0.5 + torch.rand(len(R), 1, generator=g, dtype=torch.float64) * 2
    mix = torch.randn(len(R), 1, generator=g, dtype=torch.float64) * 0.7
    d6 = torch.cat([R[:, 0] * s1, R[:, 1] * s2 + mix * R[:, 0]], dim=-1)
    return d6.float(), R


def _check_axis_angle(aa, R_true, tol=1e-6):
    """Rodrigues(aa) equals the Gram-Schmidt matrix (entries to ``tol`` in fp32 arithmetic of the producer)."""
    R_back = mo.rodrigues(aa.double())
    err = (R_back - R_true).abs().amax(dim=(-1, -2))
    assert float(err.max()) < tol, (float(err.max()), int(err.argmax()))
    ang = torch.linalg.norm(aa.double(), dim=-1)
    # (not necessarily the principal value: the quaternion's sign is not standardised in pytorch3d 0.7.2's chain, so an
    # angle t may come back as 2 pi - t about the opposite axis -- the same rotation, which is what is asserted above)
    assert float(ang.max()) <= 2 * math.pi + 1e-5


def test_rot6d_to_axis_angle_is_pinned_to_rodrigues_cpu():
    d6, R = _rot6d_cases()
    Rgs = po.rotation_6d_to_matrix(d6)
    assert float((Rgs.double() - R).abs().max()) < 5e-7          # Gram-Schmidt recovers the rotation the rows came from
    q = po.matrix_to_quaternion(Rgs)
    assert float((q.norm(dim=-1) - 1).abs().max()) < 1e-6
    # every branch of the candidate selection occurs
    m = Rgs
    qa = torch.stack([1 + m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2], 1 + m[:, 0, 0] - m[:, 1, 1] - m[:, 2, 2],
                      1 - m[:, 0, 0] + m[:, 1, 1] - m[:, 2, 2], 1 - m[:, 0, 0] - m[:, 1, 1] + m[:, 2, 2]], -1)
    assert set(qa.argmax(-1).tolist()) == {0, 1, 2, 3}
    _check_axis_angle(po.matrix_to_axis_angle(Rgs), R, tol=2e-6)


def test_rodrigues_is_a_rotation_about_its_axis():
    g = torch.Generator().manual_seed(1)
    aa = torch.randn(64, 3, generator=g, dtype=torch.float64)
    R = mo.rodrigues(aa)
    eye = torch.eye(3, dtype=torch.float64)
    assert float((R @ R.transpose(-1, -2) - eye).abs().max()) < 1e-12 and float((torch.linalg.det(R) - 1).abs().max()) < 1e-12
    assert float((torch.einsum("nij,nj->ni", R, aa) - aa).abs().max()) < 1e-12          # the axis is fixed
    assert float((R.diagonal(dim1=-1, dim2=-2).sum(-1) - (1 + 2 * torch.cos(aa.norm(dim=-1)))).abs().max()) < 1e-12


def _assets():
    import poem_v2_amd as pk
    return pk.mano.synthetic_mano_assets(3)


def _mano_properties(run, tol):
    """``run(pose (B,48), betas (B,10), center_idx) -> verts, joints`` (fp64 tensors)."""
    a = _assets()
    vt = torch.from_numpy(a["v_template"]).double()
    z48, z10 = torch.zeros(1, 48), torch.zeros(1, 10)
    # (i) zero pose, zero shape, uncentred: the template itself; joints = regressor . template and the tip vertices
    v, j = run(z48, z10, -1)
    assert float((v[0] - vt).abs().max()) < tol
    jr = torch.from_numpy(a["J_regressor"]).double() @ vt
    j21 = torch.cat([jr, vt[list(mo.TIPS)]])[list(mo.ORDER)]
    assert float((j[0] - j21).abs().max()) < tol
    # (ii) centring subtracts joint center_idx from both
    vc, jc = run(z48, z10, 9)
    assert float((vc[0] - (vt - j21[9])).abs().max()) < tol and float(jc[0, 9].abs().max()) < tol
    # (iii) shape only: v = template + shapedirs . betas
    g = torch.Generator().manual_seed(5)
    betas = torch.randn(2, 10, generator=g)
    v, _ = run(torch.zeros(2, 48), betas, -1)
    vs = vt[None] + torch.einsum("vcb,nb->nvc", torch.from_numpy(a["shapedirs"]).double(), betas.double())
    assert float((v - vs).abs().max()) < tol
    # (iv) a pure ROOT rotation moves the whole shaped mesh rigidly about the root joint (no pose-corrective term: R_0 is
    # not part of the pose feature; every descendant transform is G_0 . (rest chain))
    pose = torch.zeros(2, 48)
    pose[:, :3] = torch.tensor([[0.3, -0.8, 0.5], [2.0, 1.0, -2.2]])
    v, j = run(pose, betas, -1)
    R0 = mo.rodrigues(pose[:, :3].double())
    J0 = torch.einsum("jv,nvc->njc", torch.from_numpy(a["J_regressor"]).double(), vs)[:, 0]
    rigid = torch.einsum("nij,nvj->nvi", R0, vs - J0[:, None]) + J0[:, None]
    assert float((v - rigid).abs().max()) < tol
    # (v) a vertex bound to a single joint follows that joint's chain only: bend one finger, vertices whose weights live on
    # other fingers / the root do not move
    pose = torch.zeros(1, 48)
    pose[0, 3 * 2:3 * 2 + 3] = torch.tensor([0.0, 0.0, 0.9])          # joint 2 (index finger, second joint)
    w = a["weights"]
    off_chain = np.nonzero(w[:, [2, 3]].sum(1) == 0)[0]
    a2 = dict(a, posedirs=np.zeros_like(a["posedirs"]))
    v, _ = run(pose, z10, -1, assets=a2)
    assert len(off_chain) > 100 and float((v[0, off_chain] - vt[off_chain]).abs().max()) < tol
    on_chain = np.nonzero(w[:, [2, 3]].sum(1) > 0.5)[0]
    assert float((v[0, on_chain] - vt[on_chain]).abs().max()) > 1e-3


def test_mano_oracle_satisfies_the_models_properties():
    def run(pose, betas, c, assets=None):
        return mo.mano_lbs(assets or _assets(), pose, betas, center_idx=c)
    _mano_properties(run, 2e-8)          # fp32 skinning weights sum to 1 within 6e-8


# ---- GPU ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_rot6d_kernel_is_pinned_to_rodrigues():
    from poem_v2_amd import hip
    d6, R = _rot6d_cases()
    n = (len(d6) // 16) * 16
    d6, R = d6[:n], R[:n]
    par = torch.zeros(n // 16, 106)
    par[:, :96] = d6.reshape(-1, 96)
    par[:, 96:] = torch.arange(10.0)
    pose, betas = hip.rot6d_to_axis_angle(par.to(DEV))
    assert torch.equal(betas.cpu(), par[:, 96:])
    aa = pose.cpu().reshape(-1, 3)
    _check_axis_angle(aa, R, tol=2e-6)
    # and the oracle's restatement of the same chain agrees (what the *mano fixtures carry)
    ref = po.matrix_to_axis_angle(po.rotation_6d_to_matrix(d6))
    near_pi = (torch.linalg.norm(ref, dim=-1) - math.pi).abs() < 2e-3   # the axis sign may legitimately flip at pi
    assert float((aa - ref)[~near_pi].abs().max()) < 2e-5


@pytest.mark.gpu
def test_mano_kernel_properties_and_oracle():
    import poem_v2_amd as pk

    def run(pose, betas, c, assets=None):
        layer = pk.ManoLayer(assets or _assets(), center_idx=c, device=DEV)
        out = layer(pose.to(DEV), betas.to(DEV))
        return out.verts.double().cpu(), out.joints.double().cpu()
    _mano_properties(run, 2e-6)
    g = torch.Generator().manual_seed(9)
    pose, betas = 0.6 * torch.randn(32, 48, generator=g), torch.randn(32, 10, generator=g)
    v, j = run(pose, betas, 9)
    vo, jo = mo.mano_lbs(_assets(), pose, betas, center_idx=9)
    assert float((v - vo).abs().max()) < 2e-6 and float((j - jo).abs().max()) < 2e-6      # metres
    one = run(pose[7:8], betas[7:8], 9)
    assert torch.equal(one[0][0], v[7]) and torch.equal(one[1][0], j[7])                      # batch independence
    mpvpe = float(torch.linalg.norm(v - vo, dim=-1).mean())
    assert mpvpe <= 2e-7, mpvpe                                                               # (the path's bar is 1e-6 m)
    # every centre: the 16 kinematic joints and the five finger tips (a skinned vertex: the re-centring launch), and none
    vu, ju = run(pose, betas, -1)
    for c in range(21):
        vc, jc = run(pose, betas, c)
        assert float(jc[:, c].abs().max()) == 0.0, c                                          # the centre joint is exactly 0
        assert float((vc - (vu - ju[:, c:c + 1])).abs().max()) < 2e-7 and float((jc - (ju - ju[:, c:c + 1])).abs().max()) < 2e-7, c
        voc, joc = mo.mano_lbs(_assets(), pose, betas, center_idx=c)
        assert float((vc - voc).abs().max()) < 2e-6 and float((jc - joc).abs().max()) < 2e-6, c
    layer = pk.ManoLayer(_assets(), device=DEV)
    t = layer.zero_pose_template()
    assert t.shape == (799, 3) and float(t[9].abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        layer(torch.zeros(1, 48), torch.zeros(1, 10))                                         # CPU tensors: no fallback


@pytest.mark.gpu
def test_decoder_forward_parametric_returns_xyz_pose_shape():
    """PtEmbedTRv4.forward with PARAMETRIC_OUTPUT (ptEmb_transformer.py:371-376, pt_metro_transformer.py:139-151 upstream):
    returns (xyz stack, pose, shape) with the last layer's rows replaced by the MANO layer's joints / vertices."""
    import poem_v2_amd as pk
    from util import batch_to, build_hip_head, case_setup
    spec = dict(embed=128, nsample=4096, views=[2, 3], seed=3, parametric=True)
    cfg, w, consts, batch = case_setup(spec)
    head = build_hip_head(spec, DEV)
    layer = pk.ManoLayer(_assets(), center_idx=9, device=DEV)
    head.set_mano_layer(layer)
    tr = head.transformer
    g = torch.Generator().manual_seed(2)
    qxyz = (0.5 * torch.randn(2, 799, 3, generator=g)).to(DEV)
    qf, pxyz, pf = torch.randn(2, 799, 128, generator=g).to(DEV), torch.randn(2, 4096, 3, generator=g).to(DEV), torch.randn(2, 4096, 128, generator=g).to(DEV)
    with torch.no_grad():
        xyz, pose, shape = tr(qxyz, qf, pxyz, pf)
    assert xyz.shape == (3, 2, 799, 3) and pose.shape == (2, 48) and shape.shape == (2, 10)
    m = layer(pose, shape)
    assert torch.equal(xyz[-1, :, 21:], m.verts) and torch.equal(xyz[-1, :, :21], m.joints)
    # the head path on the same layer: last layer = MANO output + centre, earlier layers = decoder coordinates * r + centre
    feat, metas, rj = batch_to(batch, DEV)
    with torch.no_grad():
        res = head(feat, metas, rj)
    m2 = layer(res["pred_pose"].reshape(2, 48), res["pred_shape"])
    centre = rj[:, 9:10]
    assert float((res["all_coords_preds"][-1, :, 21:] - (m2.verts + centre)).abs().max()) < 1e-6
