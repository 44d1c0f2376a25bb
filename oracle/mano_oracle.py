"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the MANO layer of the medium_MANO tail.  Nothing in the product
path imports this.

PARITY UNPINNED: the reference calls ``manotorch.manolayer.ManoLayer`` (docs/installation.md:41-45 pins manotorch @v0.0.2;
call sites lib/models/bricks/pt_metro_transformer.py:120-124,147-148, lib/models/heads/ptEmb_head.py:732-736,886-892), a
third-party package that is absent from /root/reference and from this image, with licence-gated assets, and the reference
holds no test or golden vector for it.  This file restates the published MANO model (Romero, Tzionas, Black 2017: shape
blend shapes, pose-corrective blend shapes of (R - I), joint regression, forward kinematics on the 16-joint tree, linear
blend skinning) from scratch in torch fp64/fp32, in the matrix form of the paper -- deliberately NOT the evaluation order of
csrc/mano.hip, so that agreement between the two is evidence for both.  Pinned instead by properties the model defines
(tests/test_mano.py): identity pose returns the shaped template; a pure root rotation rotates the mesh rigidly about the
root joint; skinning is affine-invariant in the weights; Rodrigues of an axis-angle is a rotation about that axis by that
angle."""
import numpy as np
import torch

PARENTS = (-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14)
TIPS = (745, 317, 444, 556, 673)
ORDER = (0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20)


def rodrigues(aa):
    """(...,3) axis-angle -> (...,3,3): R = I + sin(t) K + (1 - cos(t)) K^2, K the cross-product matrix of the unit axis."""
    t = torch.linalg.norm(aa, dim=-1, keepdim=True).clamp_min(1e-30)
    k = aa / t
    K = torch.zeros(aa.shape[:-1] + (3, 3), dtype=aa.dtype)
    K[..., 0, 1], K[..., 0, 2], K[..., 1, 0] = -k[..., 2], k[..., 1], k[..., 2]
    K[..., 1, 2], K[..., 2, 0], K[..., 2, 1] = -k[..., 0], -k[..., 1], k[..., 0]
    t = t[..., None]
    eye = torch.eye(3, dtype=aa.dtype).expand(K.shape)
    return eye + torch.sin(t) * K + (1 - torch.cos(t)) * (K @ K)


def mano_lbs(assets, pose_aa, betas, center_idx=9, dtype=torch.float64):
    """assets: dict of arrays (v_template (778,3), shapedirs (778,3,10), posedirs (778,3,135), J_regressor (16,778),
    weights (778,16)); pose_aa (B,48), betas (B,10) -> verts (B,778,3), joints (B,21,3), evaluated in ``dtype``."""
    a = {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in assets.items()}
    pose, betas = pose_aa.to(dtype).reshape(-1, 16, 3), betas.to(dtype)
    B = pose.shape[0]
    R = rodrigues(pose)                                                      # (B,16,3,3)
    v_shaped = a["v_template"][None] + torch.einsum("vcb,nb->nvc", a["shapedirs"], betas)
    J = torch.einsum("jv,nvc->njc", a["J_regressor"], v_shaped)
    pose_feat = (R[:, 1:] - torch.eye(3, dtype=dtype)).reshape(B, 135)
    v_posed = v_shaped + torch.einsum("vck,nk->nvc", a["posedirs"], pose_feat)
    # forward kinematics with homogeneous 4x4 matrices, then remove the rest pose: A_j = G_j . T(-J_j)
    G = [None] * 16
    for j in range(16):
        M = torch.eye(4, dtype=dtype).repeat(B, 1, 1)
        M[:, :3, :3] = R[:, j]
        M[:, :3, 3] = J[:, j] - (J[:, PARENTS[j]] if PARENTS[j] >= 0 else 0)
        G[j] = M if PARENTS[j] < 0 else G[PARENTS[j]] @ M
    G = torch.stack(G, 1)                                                    # (B,16,4,4)
    back = torch.eye(4, dtype=dtype).repeat(B, 16, 1, 1)
    back[..., :3, 3] = -J
    A = G @ back
    vh = torch.cat([v_posed, torch.ones(B, v_posed.shape[1], 1, dtype=dtype)], -1)
    per_joint = torch.einsum("njrc,nvc->nvjr", A, vh)                        # every joint's transform applied to every vertex
    verts = torch.einsum("vj,nvjr->nvr", a["weights"], per_joint)[..., :3]   # ... blended by the skinning weights
    j21 = torch.cat([G[:, :, :3, 3], verts[:, list(TIPS)]], 1)[:, list(ORDER)]
    if center_idx is not None and center_idx >= 0:
        c = j21[:, center_idx:center_idx + 1]
        verts, j21 = verts - c, j21 - c
    return verts, j21
