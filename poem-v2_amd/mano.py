"""MANO linear blend skinning on the MI355X -- the ``ManoLayer`` the reference's parametric tail and head call
(lib/models/bricks/pt_metro_transformer.py:120-124,147-148; lib/models/heads/ptEmb_head.py:732-736,886-892 upstream:
``manotorch.manolayer.ManoLayer(joint_rot_mode="axisang", use_pca=False, mano_assets_root="assets/mano_v1_2",
center_idx=9, flat_hand_mean=True)``).

The MANO assets (``MANO_RIGHT.pkl``: ``v_template``, ``shapedirs``, ``posedirs``, ``J_regressor``, ``weights``) are
licence-gated and absent from the reference tree, and manotorch itself is a third-party dependency that is not vendored:
the layer takes the five arrays as INPUTS (:meth:`ManoLayer.from_arrays`; on a licensed machine: the fields of the MANO
pickle) and the arithmetic is restated from the published model -- parity unpinned (csrc/mano.hip).  Tests and the bench
use :func:`synthetic_mano_assets`.  No CPU fallback."""
from collections import namedtuple

import numpy as np
import torch

from . import hip

MANOOutput = namedtuple("MANOOutput", ["verts", "joints"])          # the two fields of manotorch's output the path reads
NV, NJ = 778, 16
TIP_VERTS = (745, 317, 444, 556, 673)                                # manotorch's finger-tip vertices (thumb .. pinky order of the layer)
PARENTS = (-1, 0, 1, 2, 0, 4, 5, 0, 7, 8, 0, 10, 11, 0, 13, 14)      # MANO kinematic tree


def synthetic_mano_assets(seed=0):
    """A seeded asset set with MANO's shapes and the structure the arithmetic relies on (skinning weights = convex
    combinations concentrated on 1-3 joints, regressor rows = convex combinations of vertices, small blend shapes).
    Values are NOT MANO's."""
    g = np.random.default_rng(seed)
    v = g.uniform(-0.08, 0.08, size=(NV, 3))
    shapedirs = 0.004 * g.normal(size=(NV, 3, 10))
    posedirs = 0.002 * g.normal(size=(NV, 3, 135))
    jr = g.random((NJ, NV)) ** 8
    jr /= jr.sum(1, keepdims=True)
    w = np.zeros((NV, NJ))
    for i in range(NV):
        js = g.choice(NJ, size=3, replace=False)
        w[i, js] = g.dirichlet([4.0, 1.0, 0.5])
    return {"v_template": v.astype(np.float32), "shapedirs": shapedirs.astype(np.float32),
            "posedirs": posedirs.astype(np.float32), "J_regressor": jr.astype(np.float32), "weights": w.astype(np.float32)}


class ManoLayer(torch.nn.Module):
    """``layer(pose_aa (B,48), betas (B,10)) -> MANOOutput(verts (B,778,3), joints (B,21,3))`` on the device, one launch over
    (13 vertex tiles x B) blocks.  The five asset arrays are re-laid once, here, into the kernel's table (``th_table``:
    coefficient-major blend shapes, joint regression composed with template and shape basis, joint-major skinning weights --
    csrc/mano.hip).  A head / transformer given this layer (``set_mano_layer``) attaches the table to its engine and runs the
    layer INSIDE its forward's launch graph."""

    def __init__(self, assets, center_idx=9, device="cuda:0"):
        super().__init__()
        shapes = {"v_template": (NV, 3), "shapedirs": (NV, 3, 10), "posedirs": (NV, 3, 135), "J_regressor": (NJ, NV),
                  "weights": (NV, NJ)}
        for k, shp in shapes.items():
            a = torch.as_tensor(np.asarray(assets[k]), dtype=torch.float32)
            if k == "posedirs" and tuple(a.shape) == (135, NV * 3):          # manotorch stores th_posedirs this way round
                a = a.t().reshape(NV, 3, 135)
            if tuple(a.shape) != shp:
                raise ValueError(f"MANO asset {k}: shape {tuple(a.shape)} != {shp}")
            self.register_buffer("th_" + k, a.contiguous().to(device), persistent=False)
        self.center_idx = -1 if center_idx is None else int(center_idx)
        dev = self.th_v_template.device
        if dev.type != "cuda":
            raise RuntimeError("ManoLayer runs on the MI355X HIP path only (no CPU fallback)")
        table = torch.empty(hip.lib().poem_mano_table_bytes() // 4, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            hip.check(hip.lib().poem_mano_prepare(hip.ptr(self.th_v_template), hip.ptr(self.th_shapedirs), hip.ptr(self.th_posedirs),
                                                  hip.ptr(self.th_J_regressor), hip.ptr(self.th_weights), table.data_ptr(),
                                                  hip.stream()), "poem_mano_prepare")
        self.register_buffer("th_table", table, persistent=False)

    @classmethod
    def from_arrays(cls, v_template, shapedirs, posedirs, J_regressor, weights, **kw):
        return cls(dict(v_template=v_template, shapedirs=shapedirs, posedirs=posedirs, J_regressor=J_regressor,
                        weights=weights), **kw)

    def forward(self, pose_coeffs, betas=None, **kwargs):
        if not pose_coeffs.is_cuda:
            raise RuntimeError("ManoLayer runs on the MI355X HIP path only (no CPU fallback)")
        dev = pose_coeffs.device
        B = pose_coeffs.shape[0]
        pose = pose_coeffs.detach().reshape(B, 48).to(torch.float32).contiguous()
        bet = (torch.zeros(B, 10, device=dev) if betas is None else betas.detach().reshape(B, 10)).to(torch.float32).contiguous()
        verts = torch.empty(B, NV, 3, dtype=torch.float32, device=dev)
        joints = torch.empty(B, 21, 3, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            hip.check(hip.lib().poem_mano_lbs(hip.ptr(pose), hip.ptr(bet), hip.ptr(self.th_table), B, self.center_idx,
                                              hip.ptr(verts), hip.ptr(joints), hip.stream()), "poem_mano_lbs")
        return MANOOutput(verts=verts, joints=joints)

    def zero_pose_template(self):
        """(799,3): joints then vertices of the zero-pose, zero-shape hand, centred -- what the head takes as its query
        template (ptEmb_head.py:886-892 upstream)."""
        dev = self.th_v_template.device
        out = self(torch.zeros(1, 48, device=dev), torch.zeros(1, 10, device=dev))
        return torch.cat([out.joints, out.verts], dim=1)[0]
