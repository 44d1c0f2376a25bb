"""DLT triangulation of the 21 joints from per-view 2-D predictions -- the stage that produces the head's
``reference_joints`` (lib/models/POEM.py:284-299 upstream).  MI355X-native: one ``poem_triangulate_dlt`` launch for the
whole ragged batch instead of a Python loop of per-sample ``torch.linalg.svd`` calls.  No CPU fallback.

``batch_triangulate_dlt_torch`` keeps the reference function's name, argument meaning and result
(lib/utils/triangulation.py:5-45); ``triangulate_reference_joints`` is the ragged, path-level form."""
import numpy as np
import torch

from . import hip


_OFFS_CACHE = {}


def _offsets(views, device):
    """(B+1,) int32 view offsets on the device.  A repeating layout re-uses its device tensor; a new one is staged through
    pinned memory and copied without blocking (a pageable H2D copy would hold the host until the stream has drained, i.e.
    until the previous forward has finished -- the stall the head's own layout upload avoids, csrc/forward.cpp)."""
    key = (tuple(int(v) for v in views), str(device))
    hit = _OFFS_CACHE.get(key)
    if hit is None:
        offs = np.concatenate([[0], np.cumsum(np.asarray(key[0], dtype=np.int64))]).astype(np.int32)
        hit = torch.from_numpy(offs).pin_memory().to(device, non_blocking=True)
        if len(_OFFS_CACHE) >= 64:
            _OFFS_CACHE.pop(next(iter(_OFFS_CACHE)))
        _OFFS_CACHE[key] = hit
    return hit


def triangulate_reference_joints(uv, cam_intr, cam_extr, cam_view_num):
    """uv (BN,J,2) pixel coordinates per view, cam_intr (BN,3,3), cam_extr (BN,4,4) camera->master (the batch's
    ``target_cam_extr``), cam_view_num (B,) views per sample -> (B,J,3) joints in the master frame."""
    if not uv.is_cuda:
        raise RuntimeError("triangulate_reference_joints runs on the MI355X HIP path only (no CPU fallback)")
    views = [int(v) for v in cam_view_num]
    BN, J = uv.shape[0], uv.shape[1]
    if sum(views) != BN or min(views) < 2:
        raise ValueError("cam_view_num must sum to the number of views and every sample needs >= 2 views for DLT")
    f32 = lambda t: t.to(device=uv.device, dtype=torch.float32).contiguous()   # noqa: E731
    uv, cam_intr, cam_extr = f32(uv), f32(cam_intr), f32(cam_extr)
    out = torch.empty(len(views), J, 3, dtype=torch.float32, device=uv.device)
    offs = _offsets(views, uv.device)
    hip.check(hip.lib().poem_triangulate_dlt(hip.ptr(uv), hip.ptr(cam_intr), hip.ptr(cam_extr), offs.data_ptr(), len(views),
                                             J, 1, hip.ptr(out), hip.stream()), "poem_triangulate_dlt")
    return out


def batch_triangulate_dlt_torch(kp2ds, Ks, Extrs):
    """kp2ds (B,N,J,2), Ks (B,N,3,3), Extrs (B,N,4,4) master->camera (used as is, like upstream) -> (B,J,3)."""
    if not kp2ds.is_cuda:
        raise RuntimeError("batch_triangulate_dlt_torch runs on the MI355X HIP path only (no CPU fallback)")
    B, N, J = kp2ds.shape[0], kp2ds.shape[1], kp2ds.shape[2]
    f32 = lambda t: t.to(dtype=torch.float32).contiguous()   # noqa: E731
    uv, K, T = f32(kp2ds).view(B * N, J, 2), f32(Ks).view(B * N, 3, 3), f32(Extrs).view(B * N, 4, 4)
    out = torch.empty(B, J, 3, dtype=torch.float32, device=kp2ds.device)
    offs = _offsets([N] * B, kp2ds.device)
    hip.check(hip.lib().poem_triangulate_dlt(hip.ptr(uv), hip.ptr(K), hip.ptr(T), offs.data_ptr(), B, J, 0, hip.ptr(out),
                                             hip.stream()), "poem_triangulate_dlt")
    return out


def heatmap_to_uv(uv_hmap, img_w, img_h):
    """uv_hmap (BN,J,Hh,Wh) sigmoid heat maps -> (BN,J,2) pixel coordinates: the read-out at the end of the
    reference's ``heatmap_stage`` (lib/models/POEM.py:213-222 upstream; integral_heatmap2d, integal_pose.py:194-218)."""
    if not uv_hmap.is_cuda:
        raise RuntimeError("heatmap_to_uv runs on the MI355X HIP path only (no CPU fallback)")
    h = uv_hmap.to(dtype=torch.float32).contiguous()
    BN, J, Hh, Wh = h.shape
    uv = torch.empty(BN, J, 2, dtype=torch.float32, device=h.device)
    hip.check(hip.lib().poem_heatmap_uv(hip.ptr(h), hip.ptr(uv), BN, J, Hh, Wh, float(img_w), float(img_h), hip.stream()),
              "poem_heatmap_uv")
    return uv


def reference_joints_from_heatmaps(uv_hmap, cam_intr, cam_extr, cam_view_num, img_w, img_h):
    """Heat maps of every view -> per-view 2-D joints -> ragged DLT -> (B,J,3) reference joints: the two launches that
    replace POEM.py:213-222 + :284-299 upstream."""
    return triangulate_reference_joints(heatmap_to_uv(uv_hmap, img_w, img_h), cam_intr, cam_extr, cam_view_num)
