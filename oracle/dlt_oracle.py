"""CPU oracle for the DLT triangulation stage.  TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

Restates ``batch_triangulate_dlt_torch`` (lib/utils/triangulation.py:5-45 upstream) and the ragged per-sample loop
that calls it (lib/models/POEM.py:284-299).  PINNED: tests/golden/dlt.npz holds outputs of the reference's own
function on seeded inputs (tests/golden/make_golden.py::run_dlt)."""
import numpy as np
import torch


def batch_triangulate_dlt(kp2ds, Ks, Extrs):
    """kp2ds (B,N,J,2), Ks (B,N,3,3), Extrs (B,N,4,4) (used as projection extrinsics as is) -> (B,J,3).
    triangulation.py:26-45: M = K P; rows u*M[2]-M[0], v*M[2]-M[1]; SVD; X = VT[-1,:3] / (VT[-1,3] + 1e-7)."""
    B, N, J = kp2ds.shape[0], kp2ds.shape[1], kp2ds.shape[2]
    M = torch.matmul(Ks, Extrs[..., :3, :])                                   # (B,N,3,4)      :31-32
    M = M[:, None].expand(B, J, N, 3, 4).reshape(B * J, N, 3, 4)              #                :33-34
    uv = kp2ds.permute(0, 2, 1, 3).reshape(B * J, N, 2)[..., None]            # (BJ,N,2,1)     :38
    A = (uv * M[..., 2:3, :] - M[..., :2, :]).reshape(B * J, 2 * N, 4)        #                :39-41
    VT = torch.linalg.svd(A)[2]                                               #                :43
    X = VT[:, -1, :3] / (VT[:, -1, 3:] + 1e-7)                                #                :44
    return X.reshape(B, J, 3)


def triangulate_reference_joints(uv, cam_intr, cam_extr, cam_view_num):
    """The ragged loop of POEM.py:284-299: T = inv(cam_extr) per view, one DLT per sample over its own views."""
    T = torch.linalg.inv(cam_extr)                                            # POEM.py:286
    offs = np.concatenate([[0], np.cumsum(np.asarray(cam_view_num, dtype=np.int64))])
    out = []
    for i in range(len(cam_view_num)):
        s, e = int(offs[i]), int(offs[i + 1])
        out.append(batch_triangulate_dlt(uv[s:e][None], cam_intr[s:e][None], T[s:e][None]))
    return torch.cat(out, 0)


def heatmap_to_uv(uv_hmap, img_w, img_h):
    """Tail of heatmap_stage (POEM.py:213-222) with integral_heatmap2d (integal_pose.py:194-218)."""
    BN, J, Hh, Wh = uv_hmap.shape
    pdf = uv_hmap.reshape(BN, J, -1)
    pdf = pdf / (pdf.sum(dim=-1, keepdim=True) + 1e-6)                        # POEM.py:216
    pdf = pdf.contiguous().view(BN, J, Hh, Wh)
    v_accu, u_accu = torch.sum(pdf, dim=3), torch.sum(pdf, dim=2)             # integal_pose.py:206-207
    wv = torch.arange(Hh, dtype=pdf.dtype) / Hh
    wu = torch.arange(Wh, dtype=pdf.dtype) / Wh
    v_ = torch.sum(v_accu.mul(wv), dim=-1, keepdim=True)
    u_ = torch.sum(u_accu.mul(wu), dim=-1, keepdim=True)
    uv = torch.cat([u_, v_], dim=-1)
    return torch.einsum("bij,j->bij", uv, torch.tensor([img_w, img_h], dtype=uv.dtype))   # POEM.py:219-221
