"""TEST INFRASTRUCTURE ONLY -- CPU oracle for SURVEY 8f row N4 (input side).  Nothing in the product path imports this.

Restates, in numpy:
  * the label arithmetic of ``SimpleTransform2D.__call__`` / ``SimpleTransform3DMultiView.__call__`` (lib/utils/transform.py
    :105-196, :240-281; helpers :618-705) and of ``MultiviewWebDataset.process_data_item`` (lib/data_wds/multiview_wds.py
    :62-145) and ``collation_random_n_views`` (lib/utils/collation.py:7-25).  PINNED: tests/golden/transform.npz holds the
    outputs of the reference's own classes on seeded labels (tests/golden/make_golden_transform.py).
  * ``cv2.warpAffine(img, M, (W, H), flags=INTER_LINEAR, borderMode=BORDER_CONSTANT)`` for uint8 images and the
    ``tvF.to_tensor`` / ``tvF.normalize`` tail.  PARITY UNPINNED for the warp: OpenCV (requirements.txt:11 pins
    opencv-python==4.5.1.48) is a third-party dependency that is absent from /root/reference and from this image, and the
    reference has no golden images.  The function below restates OpenCV 4.5's published algorithm
    (modules/imgproc/src/imgwarp.cpp: cv::warpAffine -> WarpAffineInvoker -> remap / remapBilinear<FixedPtCast<int, uchar,
    15>>, tables from initInterTab2D): inverse matrix in fp64, 10-bit fixed-point coordinates (AB_BITS) rounded to 1/32
    pixel (INTER_BITS 5), 15-bit integer bilinear weights (INTER_REMAP_COEF_BITS), constant border 0.  What the tests pin
    independently of OpenCV: exact agreement with a float64 bilinear interpolation at the 1/32-pixel-quantised source
    coordinates (the defining property of that fixed-point scheme), identity / integer-shift / flip warps being exact
    copies, and ``to_tensor`` / ``normalize`` against torch on all 256 byte values.
"""
import pickle
import math
import random

import numpy as np

AB_BITS, INTER_BITS, COEF_BITS = 10, 5, 15
NUM_JOINTS = 21


# ---- cv2.warpAffine, uint8, INTER_LINEAR, BORDER_CONSTANT(0) ----------------------------------------------------------
def invert_2x3(M):
    """cv::warpAffine's matrix preparation (imgwarp.cpp, `if( !(flags & WARP_INVERSE_MAP) )`)."""
    m = np.array(M, dtype=np.float64).reshape(2, 3).copy().reshape(-1)
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    m[0] = A11
    m[1] *= -D
    m[3] *= -D
    m[4] = A22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


def fixed_point_coords(M, W, H):
    """-> integer source coordinates (sx, sy) and 5-bit fractions (ax, ay) of every destination pixel, (H, W) each."""
    m = invert_2x3(M)
    x = np.arange(W, dtype=np.float64)
    y = np.arange(H, dtype=np.float64)
    scale = float(1 << AB_BITS)
    adelta = np.rint(m[0] * x * scale).astype(np.int64)                      # saturate_cast<int>: round half to even
    bdelta = np.rint(m[3] * x * scale).astype(np.int64)
    round_delta = (1 << AB_BITS) // (1 << INTER_BITS) // 2
    X0 = np.rint((m[1] * y + m[2]) * scale).astype(np.int64) + round_delta
    Y0 = np.rint((m[4] * y + m[5]) * scale).astype(np.int64) + round_delta
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx = np.clip(X >> INTER_BITS, -32768, 32767)                             # saturate_cast<short>
    sy = np.clip(Y >> INTER_BITS, -32768, 32767)
    return sx, sy, X & 31, Y & 31


def warp_affine_u8(img, M, dsize):
    """img uint8 (h, w, C); M (2,3) source->destination; dsize = (W, H) -> uint8 (H, W, C)."""
    W, H = int(dsize[0]), int(dsize[1])
    h, w = img.shape[:2]
    sx, sy, ax, ay = fixed_point_coords(M, W, H)
    acc = np.zeros((H, W, img.shape[2]), dtype=np.int64)
    for dy, wy in ((0, 32 - ay), (1, ay)):
        for dx, wx in ((0, 32 - ax), (1, ax)):
            yy, xx = sy + dy, sx + dx
            inside = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            tap = np.where(inside[..., None], img[np.clip(yy, 0, h - 1), np.clip(xx, 0, w - 1)].astype(np.int64), 0)
            acc += (wy * wx * 32)[..., None] * tap                          # (32-a)(32-b) 2^-10 * 2^15
    return ((acc + (1 << (COEF_BITS - 1))) >> COEF_BITS).astype(np.uint8)


def color_jitter_u8(img, gains):
    """transform.py:159-164: `image[:, :, c] = np.clip(image[:, :, c] * g, 0, 255)` on a uint8 array (C cast, truncation)."""
    out = img.copy()
    for c in range(3):
        out[:, :, c] = np.clip(out[:, :, c] * float(gains[c]), 0, 255)
    return out


def to_tensor_normalize(img_u8):
    """tvF.to_tensor (HWC uint8 -> CHW fp32 / 255) then tvF.normalize(mean 0.5, std 1)   [transform.py:166-169]."""
    t = np.ascontiguousarray(img_u8.transpose(2, 0, 1)).astype(np.float32) / np.float32(255)
    return (t - np.float32(0.5)) / np.float32(1)


# ---- label arithmetic ---------------------------------------------------------------------------------------------------
def rotation_matrix(rot, size=3):                                            # :618-634
    r = np.eye(size, dtype=np.float32)
    if rot != 0:
        sn, cs = np.sin(rot), np.cos(rot)
        r[0, :2] = [cs, -sn]
        r[1, :2] = [sn, cs]
    return r


def affine_no_rot(center, scale, res):                                       # :697-705
    a = np.zeros((3, 3))
    sr = float(res[0]) / float(res[1])
    a[0, 0] = float(res[0]) / scale
    a[1, 1] = float(res[1]) / scale * sr
    a[0, 2] = res[0] * (-float(center[0]) / scale + 0.5)
    a[1, 2] = res[1] * (-float(center[1]) / scale * sr + 0.5)
    a[2, 2] = 1
    return a


def affine_transform(center, scale, out_res, rot=0):                         # :674-681
    R = rotation_matrix(rot)
    c = R.dot(np.concatenate([center, np.ones(1)]))[:2]
    return affine_no_rot(c, scale, out_res).dot(R).astype(np.float32)


def affine_post_rot(center, scale, optical_center, out_res, rot=0):          # :684-694
    R = rotation_matrix(rot)
    T = np.eye(3)
    T[0, 2], T[1, 2] = -optical_center[0], -optical_center[1]
    Ti = T.copy()
    Ti[:2, 2] *= -1
    c = Ti.dot(R).dot(T).dot(np.concatenate([center, np.ones(1)]))
    return affine_no_rot(c[:2], scale, out_res).astype(np.float32)


def transform_coords(pts, A):                                                # :637-646
    hom = np.concatenate([pts, np.ones([np.array(pts).shape[0], 1])], 1)
    return A.dot(hom.transpose()).transpose()[:, :2]


def random_occlusion(image, center, scale, prob):
    """RandomOcclusion.__call__ on the jittered bbox (transform.py:21-66 called from :111-116; box = center_scale_to_box,
    :1083-1101): one np.random coin; if taken, four ``random.random()`` draws (patch area <= 20 % of the box, aspect in
    [0.5, 2], position) and -- when the patch lies inside the image -- an (h, w, 3) block of uniform noise * 255 written INTO
    the caller's uint8 array (numpy truncates on assignment)."""
    if np.random.rand() > prob:
        return image
    xmin, ymin = center[0] - scale * 0.5, center[1] - scale * 0.5
    bw, bh = (xmin + scale) - xmin, (ymin + scale) - ymin
    H, W = image.shape[0], image.shape[1]
    area = (random.random() * 0.2) * bw * bh
    ratio = random.random() * (1 / 0.5 - 0.5) + 0.5
    h, w = math.sqrt(area * ratio), math.sqrt(area / ratio)
    px = random.random() * (bw - w - 1) + xmin
    py = random.random() * (bh - h - 1) + ymin
    if px >= 0 and py >= 0 and px + w < W and py + h < H:
        px, py, w, h = int(px), int(py), int(w), int(h)
        if not image.flags.writeable:          # (decoded records hand out read-only buffers; upstream's arrays are writable)
            image = image.copy()
        image[py:py + h, px:px + w, :] = np.random.rand(h, w, 3) * 255
    return image


def simple_transform_3d_multiview(image, label, out_size=(256, 256), is_train=False, aug=None, no_rot=False):
    """One view.  aug = None (evaluation: AUG false) or dict(center_jit, scale_jit, rot_jit, rot_prob, color_jit
    [, occlusion_prob]) -- draws from np.random / random exactly where upstream does; ``occlusion_prob`` absent = OCCLUSION
    False as in every released config (upstream's default when the key is missing is True / 0.1, transform.py:83-84)."""
    if aug is not None:
        c_factor = np.random.normal(loc=0, scale=aug["center_jit"], size=2)
        center = label["bbox_center"] + c_factor * label["bbox_scale"]
        s_factor = np.random.normal(loc=1, scale=aug["scale_jit"])
        scale = label["bbox_scale"] * s_factor
        r_factor = np.random.normal(loc=0, scale=aug["rot_jit"])
        rot = np.deg2rad(r_factor) if (not no_rot and np.random.rand() <= aug["rot_prob"]) else 0.0
        if aug.get("occlusion_prob") is not None:
            image = random_occlusion(image, center, scale, aug["occlusion_prob"])
    else:
        scale, center, rot = label["bbox_scale"], label["bbox_center"], 0.0
    R3 = rotation_matrix(rot)
    A = affine_transform(center, scale, out_size, rot)
    j2d = transform_coords(label["joints_2d"], A).astype(np.float32)
    if not is_train:
        vis = np.full(NUM_JOINTS, 1.0, dtype=np.float32)
    elif label["joints_vis"].sum() < NUM_JOINTS * 0.3:
        vis = np.full(NUM_JOINTS, 0.0, dtype=np.float32)
    else:
        vis = (((j2d[:, 0] >= 0) & (j2d[:, 0] < out_size[0])) & ((j2d[:, 1] >= 0) & (j2d[:, 1] < out_size[1]))).astype(np.float32)
        if vis.sum() < NUM_JOINTS * 0.3:
            vis = np.full(NUM_JOINTS, 0.0, dtype=np.float32)
    warped = warp_affine_u8(image, A[:2, :], (int(out_size[0]), int(out_size[1])))
    if aug is not None:
        lo, hi = 1 - aug["color_jit"], 1 + aug["color_jit"]
        warped = color_jitter_u8(warped, [random.uniform(lo, hi) for _ in range(3)])
    intr = label["cam_intr"]
    Apost = affine_post_rot(center, scale, np.array([intr[0, 2], intr[1, 2]]), out_size, rot)
    return {"rot_rad": rot, "rot_mat3d": R3, "affine": A, "image": to_tensor_normalize(warped), "image_u8": warped,
            "target_bbox_center": center, "target_bbox_scale": scale, "target_joints_2d": j2d, "target_joints_vis": vis,
            "image_path": label["image_path"], "affine_postrot": Apost, "extr_prerot": R3,
            "target_cam_intr": Apost.dot(label["cam_intr"]),
            "target_joints_3d": R3.dot(label["joints_3d"].transpose(1, 0)).transpose(),
            "target_verts_3d": R3.dot(label["verts_3d"].transpose(1, 0)).transpose(),
            "target_joints_3d_no_rot": label["joints_3d"], "target_verts_3d_no_rot": label["verts_3d"]}


def process_data_item(item, inv_extr=False, random_n_views=False, view_range=None, **tf):
    """multiview_wds.py:62-145 for a decoded record {"__key__", "image_i.ext": uint8 HxWx3, "label.pyd": dict of lists}."""
    imgs = {k: v for k, v in item.items() if k.startswith("image")}
    ext = "png" if any("png" in k for k in imgs) else "jpg"
    n_cams = len(imgs)
    labels = dict(item["label.pyd"])
    if "mano_pose" in labels:
        labels["mano_pose"] = [labels["mano_pose"][i].reshape(-1)[:48].reshape(16, 3) for i in range(n_cams)]
    else:
        labels["mano_pose"] = [np.zeros((16, 3)) for _ in range(n_cams)]
        labels["mano_shape"] = [np.zeros(10) for _ in range(n_cams)]
    if inv_extr:
        labels["cam_extr"] = [np.linalg.inv(labels["cam_extr"][i]) for i in range(n_cams)]
    indices = list(range(n_cams))
    if random_n_views:
        random.shuffle(indices)
        n = int(round(random.gauss(4, 2)))
        n = min(max(view_range[0], n), view_range[1])
        indices = indices[:min(n, n_cams)]
    master = indices[0]
    T_master = labels["cam_extr"][master]
    res = {}
    for ind in indices:
        img = imgs[f"image_{ind}.{ext}"]
        if labels.get("request_flip", False):
            K = labels["cam_intr"][ind]
            M = np.array([[-1, 0, 2 * K[0, 2]], [0, 1, 0]], dtype=np.float32)
            img = warp_affine_u8(img, M, labels["raw_size"][ind])
        lab = {k: v[ind] for k, v in labels.items() if k not in ["request_flip"]}
        tgt = simple_transform_3d_multiview(img, lab, no_rot=ind == master, **tf)
        T = np.linalg.inv(T_master) @ lab["cam_extr"]
        pre = np.concatenate([np.concatenate([tgt["extr_prerot"], np.zeros((3, 1))], axis=1), np.array([[0, 0, 0, 1]])], axis=0)
        tgt["target_cam_extr"] = np.linalg.inv(pre @ np.linalg.inv(T)).astype(np.float32)
        tgt.update(lab)
        for k, v in tgt.items():
            res.setdefault(k, []).append(v)
    for k in res:
        if isinstance(res[k][0], (int, float, np.ndarray)):
            res[k] = np.stack(res[k])
    res["master_id"] = 0
    res["master_serial"] = labels["cam_serial"][master]
    res["master_joints_3d"] = labels["joints_3d"][master]
    res["master_verts_3d"] = labels["verts_3d"][master]
    res["__key__"] = item["__key__"]
    return res


def collation_random_n_views(batch):                                         # collation.py:7-25 (numpy instead of Tensor)
    if not isinstance(batch, list):
        batch = [batch]
    out = {}
    for k in batch[0]:
        if isinstance(batch[0][k], np.ndarray) and not isinstance(batch[0][k][0], str):
            out[k] = np.concatenate([b[k] for b in batch], axis=0).astype(np.float32)
        else:
            out[k] = [b[k] for b in batch]
    out["cam_view_num"] = np.array([b["target_joints_3d"].shape[0] for b in batch])
    return out


# ---- seeded synthetic records (shared by the tests and the golden generator) -----------------------------------------
def synthetic_frame(seed, n_cams=4, raw=(640, 480), ext="png", dtype=np.float32):
    """A decoded multi-view record of the shape the dataset tars hold (multiview_wds.py:63-75): per-camera images and a
    label dict of per-camera lists."""
    g = np.random.default_rng(seed)
    W, H = raw
    lab = {k: [] for k in ("cam_intr", "cam_extr", "cam_serial", "joints_3d", "verts_3d", "joints_2d", "joints_vis",
                           "bbox_center", "bbox_scale", "image_path", "raw_size", "mano_pose", "mano_shape", "idx")}
    item = {"__key__": f"frame{seed:06d}"}
    hand = np.array([0.0, 0.0, 0.6])
    for i in range(n_cams):
        K = np.array([[580 + 20 * g.random(), 0, W / 2 + 10 * g.normal()], [0, 580 + 20 * g.random(), H / 2 + 10 * g.normal()],
                      [0, 0, 1]], dtype)
        ang = 2 * np.pi * i / max(n_cams, 1) + 0.1 * g.normal()
        R = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
        T = np.eye(4)
        T[:3, :3] = R
        T[:3, 3] = hand - R @ hand + 0.01 * g.normal(size=3)
        j3d = (hand + 0.04 * g.normal(size=(21, 3))).astype(dtype)
        v3d = (hand + 0.04 * g.normal(size=(778, 3))).astype(dtype)
        uv = (K.astype(np.float64) @ j3d.T.astype(np.float64)).T
        j2d = (uv[:, :2] / uv[:, 2:]).astype(dtype)
        c = 0.5 * (j2d.min(0) + j2d.max(0))
        s = float(1.7 * (j2d.max(0) - j2d.min(0)).max())
        lab["cam_intr"].append(K)
        lab["cam_extr"].append(T.astype(dtype))
        lab["cam_serial"].append(f"cam{i}")
        lab["joints_3d"].append(j3d)
        lab["verts_3d"].append(v3d)
        lab["joints_2d"].append(j2d)
        lab["joints_vis"].append(np.ones(21, dtype))
        lab["bbox_center"].append(c.astype(dtype))
        lab["bbox_scale"].append(dtype(s))
        lab["image_path"].append(f"seq/{seed}/{i}.{ext}")
        lab["raw_size"].append((W, H))
        lab["mano_pose"].append((0.1 * g.normal(size=48)).astype(dtype))
        lab["mano_shape"].append((0.1 * g.normal(size=10)).astype(dtype))
        lab["idx"].append(seed * 16 + i)
        yy, xx = np.mgrid[0:H, 0:W]
        base = (np.stack([xx * 255 // max(W - 1, 1), yy * 255 // max(H - 1, 1), (xx + yy) % 256], -1)).astype(np.int64)
        img = np.clip(base + g.integers(-40, 40, size=(H, W, 3)), 0, 255).astype(np.uint8)
        item[f"image_{i}.{ext}"] = img
    item["label.pyd"] = lab
    return item


def label_bytes(lab):
    return pickle.dumps(lab, protocol=4)
