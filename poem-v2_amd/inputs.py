"""Synthetic multi-view inputs of the path's boundary: backbone feature maps, pinhole
intrinsics, camera->master extrinsics on a ring around the hand, triangulated reference joints.  Seeded and
generated on CPU so that the oracle, the HIP path and the bench see identical numbers.

Layout follows the reference's batch contract (lib/utils/collation.py:7-25): per-view tensors are concatenated
over samples -> (BN, ...) with ``cam_view_num`` = views per sample; view 0 of every sample is the master whose
extrinsic is the identity (lib/data_wds/multiview_wds.py:97-126)."""
import math

import numpy as np
import torch


def ring_extrinsics(n_views, ring=8, hand=(0.0, 0.0, 0.6), jitter=None):
    """camera->master 4x4 for views 0..n_views-1: camera n sits on a circle around ``hand`` (radius = |hand|) at
    angle 2*pi*n/ring about the y axis and looks at the hand; view 0 is the identity."""
    hand = torch.tensor(hand, dtype=torch.float64)
    out = torch.eye(4, dtype=torch.float64).repeat(n_views, 1, 1)
    for n in range(n_views):
        th = 2 * math.pi * n / ring
        if jitter is not None and n > 0:
            th = th + float(jitter[n])
        R = torch.tensor([[math.cos(th), 0, math.sin(th)], [0, 1, 0], [-math.sin(th), 0, math.cos(th)]],
                         dtype=torch.float64)
        out[n, :3, :3] = R
        out[n, :3, 3] = hand - R @ hand
    return out.float()


def synthetic_batch(views, seed=0, in_channels=160, feat_hw=16, img=256, ring=None, nan_views=None):
    """views: list of views per sample.  Returns the arguments of ``POEM_Generalized_Head.forward``.  ``nan_views``: global view
    indices whose feature maps are NaN (the *nan fixtures: what `torch.nan_to_num`, ptEmb_head.py:944 upstream, is there for)."""
    g = torch.Generator().manual_seed(seed)
    views = [int(v) for v in views]
    B, BN = len(views), int(sum(views))
    ring = ring or max(8, max(views))
    mlvl_feat = torch.randn(BN, in_channels, feat_hw, feat_hw, generator=g)
    for v in (nan_views or ()):
        mlvl_feat[int(v)] = float("nan")
    K = torch.tensor([[300.0, 0, img / 2], [0, 300.0, img / 2], [0, 0, 1]])
    cam_intr = K[None].repeat(BN, 1, 1).contiguous()
    extr = []
    for n in views:
        jit = 0.05 * torch.randn(n, generator=g)
        extr.append(ring_extrinsics(n, ring=ring, jitter=jit))
    cam_extr = torch.cat(extr, 0).contiguous()
    reference_joints = torch.tensor([0.0, 0.0, 0.6]) + 0.03 * torch.randn(B, 21, 3, generator=g)
    img_metas = {
        "inp_img_shape": (img, img),
        "cam_intr": cam_intr,
        "cam_extr": cam_extr,
        "master_id": [0] * B,
        "cam_view_num": np.asarray(views, dtype=np.int64),
    }
    return {"mlvl_feat": mlvl_feat, "img_metas": img_metas, "reference_joints": reference_joints}


def synthetic_template(seed=1234):
    """Seeded synthetic (799,3) zero-pose hand template in metres, centred at joint 9 (rows 0..20 joints, 21..798
    vertices).  MANO assets are licence-gated (docs/datasets.md:31-38 upstream); on a licensed machine pass the
    real ManoLayer output instead."""
    g = torch.Generator().manual_seed(seed)
    t = (torch.rand(799, 3, generator=g) * 2 - 1) * 0.08
    return t - t[9:10]


def synthetic_pyramid(views, seed=0):
    """HRNet-shaped multi-level features of a 256x256 image (lib/models/POEM.py:240-246 upstream):
    (BN,40,64,64), (BN,80,32,32), (BN,160,16,16), (BN,320,8,8), seeded N(0,1)."""
    g = torch.Generator().manual_seed(2000 + seed)
    return [torch.randn(views, c, r, r, generator=g) for c, r in zip((40, 80, 160, 320), (64, 32, 16, 8))]


def synthetic_images(views, seed=0, img=256):
    """(BN,3,img,img) seeded N(0, 0.3^2) "images" for the end-to-end timing scope."""
    g = torch.Generator().manual_seed(3000 + seed)
    return 0.3 * torch.randn(views, 3, img, img, generator=g)


def synthetic_frame(seed, n_cams=4, raw=(640, 480), ext="png", dtype=np.float32):
    """A decoded multi-view record of the layout the dataset tars hold (``image_<i>.<ext>`` per camera + ``label.pyd`` =
    dict of per-camera lists; lib/data_wds/multiview_wds.py:63-75 upstream), seeded: cameras on a ring around a hand at
    0.6 m, jittered intrinsics, Gaussian joints / vertices, a gradient + noise image per camera.  The dataset tars are not
    available offline; ``scripts/eval_single.py --shards`` and the tests write shards of these records instead.
    (tests/test_transform.py checks that the test infrastructure's own generator produces the identical records.)"""
    g = np.random.default_rng(seed)
    W, H = raw
    names = ("cam_intr", "cam_extr", "cam_serial", "joints_3d", "verts_3d", "joints_2d", "joints_vis", "bbox_center",
             "bbox_scale", "image_path", "raw_size", "mano_pose", "mano_shape", "idx")
    lab = {k: [] for k in names}
    item = {"__key__": f"frame{seed:06d}"}
    hand = np.array([0.0, 0.0, 0.6])
    yy, xx = np.mgrid[0:H, 0:W]
    ramp = np.stack([xx * 255 // max(W - 1, 1), yy * 255 // max(H - 1, 1), (xx + yy) % 256], -1).astype(np.int64)
    for i in range(n_cams):
        K = np.array([[580 + 20 * g.random(), 0, W / 2 + 10 * g.normal()], [0, 580 + 20 * g.random(), H / 2 + 10 * g.normal()],
                      [0, 0, 1]], dtype)
        ang = 2 * np.pi * i / max(n_cams, 1) + 0.1 * g.normal()
        T = np.eye(4)
        T[:3, :3] = [[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]]
        T[:3, 3] = hand - T[:3, :3] @ hand + 0.01 * g.normal(size=3)
        j3d = (hand + 0.04 * g.normal(size=(21, 3))).astype(dtype)
        v3d = (hand + 0.04 * g.normal(size=(778, 3))).astype(dtype)
        uvw = (K.astype(np.float64) @ j3d.T.astype(np.float64)).T
        j2d = (uvw[:, :2] / uvw[:, 2:]).astype(dtype)
        span = j2d.max(0) - j2d.min(0)
        per_cam = {"cam_intr": K, "cam_extr": T.astype(dtype), "cam_serial": f"cam{i}", "joints_3d": j3d, "verts_3d": v3d,
                   "joints_2d": j2d, "joints_vis": np.ones(21, dtype), "bbox_center": (0.5 * (j2d.min(0) + j2d.max(0))).astype(dtype),
                   "bbox_scale": dtype(float(1.7 * span.max())), "image_path": f"seq/{seed}/{i}.{ext}", "raw_size": (W, H)}
        per_cam["mano_pose"] = (0.1 * g.normal(size=48)).astype(dtype)
        per_cam["mano_shape"] = (0.1 * g.normal(size=10)).astype(dtype)
        per_cam["idx"] = seed * 16 + i
        for k in names:
            lab[k].append(per_cam[k])
        item[f"image_{i}.{ext}"] = np.clip(ramp + g.integers(-40, 40, size=(H, W, 3)), 0, 255).astype(np.uint8)
    item["label.pyd"] = lab
    return item
