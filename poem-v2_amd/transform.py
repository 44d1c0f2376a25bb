"""Input-side transform of the multi-view pipeline: ``SimpleTransform3DMultiView`` with the image
work (crop / warp / colour jitter / to-tensor / normalise) on the MI355X and the label arithmetic on the host.

Mirrors ``lib/utils/transform.py`` upstream: ``SimpleTransform2D.__call__`` (:105-196), ``SimpleTransform3DMultiView``
(:240-281) and the affine helpers (:618-705); same config keys, same result keys and dtypes, the same consumption of
``np.random`` / ``random`` (so a seeded upstream run and a seeded run here pick the same augmentation) -- but formulated per
frame: closed-form crop matrices for all views at once (:func:`crop_geometry`) instead of per-view 3x3 product chains.  The image chain of the reference is ``cv2.warpAffine`` -> colour jitter -> ``tvF.to_tensor`` ->
``tvF.normalize`` per view on the host; here every view handed to :func:`warp_views` is processed by ONE launch of
``poem_warp_affine`` (csrc/warp.hip) from one pinned upload, and ``results["image"]`` is a device tensor.

No CPU fallback: without the HIP library / a GPU the image side raises (``hip.lib()``); the label side
(:meth:`SimpleTransform3DMultiView.labels`) is pure numpy and runs anywhere.
"""
import math
import random

import numpy as np
import torch

from . import hip
from .builder import Registry, build_from_cfg

TRANSFORM = Registry("transform")                     # lib/utils/builder.py:311 upstream
NUM_JOINTS = 21                                       # CONST.NUM_JOINTS


def build_transform(cfg, **kwargs):                   # lib/utils/builder.py:335-336
    return build_from_cfg(cfg, TRANSFORM, **kwargs)


# ---- crop geometry of all views of a frame at once ------------------------------------------------------------------
# What the reference builds per view through chains of 3x3 products (lib/utils/transform.py:618-705: the in-plane
# rotation, the crop about the rotated bbox centre, the "post-rotation" crop about the centre rotated around the
# principal point) is written here in closed form over a leading view axis.  The 3x3 products upstream only ever add
# exact zeros to a single product per entry, so the closed form below is the same fp64 number, entry by entry; the
# association order of the two-term sums is kept ((a*x + b*y) + t) because the fixtures compare with ``==``.
def inplane_rotations(rot):
    """rot (V,) radians -> (V,3,3) fp32 rotations about the optical axis (identity where rot == 0)."""
    rot = np.asarray(rot, dtype=np.float64).reshape(-1)
    out = np.zeros((rot.size, 3, 3), dtype=np.float32)
    cs, sn = np.cos(rot), np.sin(rot)
    out[:, 0, 0] = out[:, 1, 1] = cs
    out[:, 0, 1], out[:, 1, 0] = -sn, sn
    out[:, 2, 2] = 1
    return out


def _crop_about(cx, cy, scale, out_size):
    """Scale + shift that maps the ``scale``-sized square centred at (cx, cy) onto the output window -> (sx, sy, tx, ty).
    Evaluated in the dtype of ``scale``: the records store the bbox scale as a numpy scalar and the reference divides plain
    Python floats by it, which (numpy >= 2 promotion) keeps fp32 labels in fp32 -- the fixtures pin that."""
    dt = scale.dtype
    w, h = dt.type(out_size[0]), dt.type(out_size[1])
    aspect = dt.type(float(out_size[0]) / float(out_size[1]))
    half = dt.type(0.5)
    cx, cy = cx.astype(dt), cy.astype(dt)
    return w / scale, h / scale * aspect, w * (-cx / scale + half), h * (-cy / scale * aspect + half)


def crop_geometry(center, scale, rot, principal, out_size):
    """center (V,2), scale (V,), rot (V,), principal point (V,2), all fp64 -> three (V,3,3) fp32 stacks:
    rotation R, the image warp  A = Crop(R c) . R  (source pixel -> output pixel, what ``cv2.warpAffine`` receives) and the
    intrinsics update  P = Crop(o + R (c - o))  (no rotation part: the rotation moves to the extrinsics)."""
    center = np.asarray(center, dtype=np.float64).reshape(-1, 2)
    scale = np.asarray(scale).reshape(-1)
    if scale.dtype not in (np.float32, np.float64):
        scale = scale.astype(np.float64)
    principal = np.asarray(principal, dtype=np.float64).reshape(-1, 2)
    R = inplane_rotations(rot)
    cs, ms, sn = (R[:, 0, 0].astype(np.float64), R[:, 0, 1].astype(np.float64), R[:, 1, 0].astype(np.float64))
    cx, cy, ox, oy = center[:, 0], center[:, 1], principal[:, 0], principal[:, 1]
    V = scale.size
    # warp: crop about the rotated centre, then the rotation folded in (columns 0/1 scale the rotation's rows)
    sx, sy, tx, ty = _crop_about(cs * cx + ms * cy, sn * cx + cs * cy, scale, out_size)
    A = np.zeros((V, 3, 3))
    A[:, 0, 0], A[:, 0, 1], A[:, 0, 2] = sx * cs, sx * ms, tx
    A[:, 1, 0], A[:, 1, 1], A[:, 1, 2] = sy * sn, sy * cs, ty
    A[:, 2, 2] = 1
    # intrinsics: the centre rotated about the principal point, o + R (c - o), with the translation column of
    # T(o) R T(-o) formed first as upstream's left-to-right product does
    shift_x = (cs * -ox + ms * -oy) + ox
    shift_y = (sn * -ox + cs * -oy) + oy
    sx, sy, tx, ty = _crop_about((cs * cx + ms * cy) + shift_x, (sn * cx + cs * cy) + shift_y, scale, out_size)
    P = np.zeros((V, 3, 3))
    P[:, 0, 0], P[:, 1, 1], P[:, 0, 2], P[:, 1, 2], P[:, 2, 2] = sx, sy, tx, ty, 1
    return R, A.astype(np.float32), P.astype(np.float32)


def _rows_times(mat, pts):
    """mat (3,3), pts (n,3) -> (n,3) rows ``mat @ p`` in the promoted dtype, as ONE matrix product ``mat.dot(pts.T).T`` -- the
    form upstream uses (transform.py:311-312), so that the host BLAS rounds both sides alike (its k-loop is fma-accumulated:
    a hand-written ``(m0 x + m1 y) + m2 z`` differs from it in the last bit on ~1 % of the coordinates)."""
    dt = np.result_type(mat.dtype, np.asarray(pts).dtype)
    return mat.astype(dt).dot(np.asarray(pts).astype(dt).transpose(1, 0)).transpose()


# ---- the device stage -------------------------------------------------------------------------------------------------
def invert_affine(m):
    """What ``cv::warpAffine`` does with the 2x3 matrix it is given (no WARP_INVERSE_MAP): widen to fp64, invert.
    Plain Python floats: every product / sum individually rounded, as in OpenCV's scalar code."""
    m = [float(v) for v in np.asarray(m, dtype=np.float64).reshape(-1)[:6]]
    d = m[0] * m[4] - m[1] * m[3]
    d = 1.0 / d if d != 0 else 0.0
    a11, a22 = m[4] * d, m[0] * d
    m[0], m[1], m[3], m[4] = a11, m[1] * -d, m[3] * -d, a22
    b1 = -m[0] * m[2] - m[1] * m[5]
    b2 = -m[3] * m[2] - m[4] * m[5]
    m[2], m[5] = b1, b2
    return m


class _Staging:
    """Two pinned host buffers per device, used alternately; an event per buffer marks the end of the upload that last
    read it, so filling never races an in-flight copy and never blocks on the stream's compute."""

    def __init__(self):
        self.buf, self.ev, self.turn = [None, None], [None, None], 0

    def take(self, nbytes):
        i = self.turn
        self.turn ^= 1
        if self.ev[i] is not None:
            self.ev[i].synchronize()
        if self.buf[i] is None or self.buf[i].numel() < nbytes:
            self.buf[i] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, pin_memory=True)
        return i, self.buf[i]


_staging = {}


def warp_views(images, affines, out_size, gains=None, device="cuda:0", out="f32"):
    """images: list of uint8 (H_i, W_i, 3) arrays (ragged sizes); affines: per view a (2,3) / (3,3) source->destination
    matrix (what the reference hands to ``cv2.warpAffine``); out_size = (W, H); gains: per view 3 colour gains or None.
    -> device tensor (V,3,H,W) fp32 = warped / 255 - 0.5  (``out="f32"``)  or  (V,H,W,3) uint8  (``out="u8"``).
    One pinned upload and one kernel launch for all views."""
    lib = hip.lib()
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("warp_views runs on the GPU only (there is no CPU fallback)")
    V = len(images)
    if V == 0 or len(affines) != V:
        raise ValueError("warp_views needs one affine per image and at least one image")
    ow, oh = int(out_size[0]), int(out_size[1])
    head = V * (8 + 8 + 48 + 24)
    head = (head + 15) & ~15
    offs, total = [], head
    for im in images:
        if im.dtype != np.uint8 or im.ndim != 3 or im.shape[2] != 3:
            raise ValueError("images must be uint8 (H, W, 3)")
        offs.append(total - head)
        total += (im.size + 15) & ~15
    st = _staging.setdefault(dev.index or 0, _Staging())
    slot, pinned = st.take(total)
    host = pinned.numpy()
    host[0:8 * V].view(np.int64)[:] = offs
    host[8 * V:16 * V].view(np.int32)[:] = np.asarray([[im.shape[0], im.shape[1]] for im in images], np.int32).reshape(-1)
    host[16 * V:64 * V].view(np.float64)[:] = np.asarray([invert_affine(np.asarray(a)[:2]) for a in affines]).reshape(-1)
    if gains is not None:
        host[64 * V:88 * V].view(np.float64)[:] = np.asarray(gains, np.float64).reshape(-1)
    for im, o in zip(images, offs):
        host[head + o:head + o + im.size] = np.ascontiguousarray(im).reshape(-1)
    with torch.cuda.device(dev):
        blob = torch.empty(total, dtype=torch.uint8, device=dev)
        blob.copy_(pinned[:total], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        st.ev[slot] = ev
        base = blob.data_ptr()
        f32 = torch.empty(V, 3, oh, ow, dtype=torch.float32, device=dev) if out == "f32" else None
        u8 = torch.empty(V, oh, ow, 3, dtype=torch.uint8, device=dev) if out == "u8" else None
        if f32 is None and u8 is None:
            raise ValueError("out must be 'f32' or 'u8'")
        hip.check(lib.poem_warp_affine(base + head, base, base + 8 * V, base + 16 * V,
                                       base + 64 * V if gains is not None else None, hip.ptr(f32) if f32 is not None else None,
                                       hip.ptr(u8, torch.uint8) if u8 is not None else None, V, oh, ow, hip.stream()), "poem_warp_affine")
        blob.record_stream(torch.cuda.current_stream())
    return f32 if f32 is not None else u8


# ---- the transform ------------------------------------------------------------------------------------------------------
class ViewDraw:
    """The random part of one view's augmentation: bbox centre / scale jitter, in-plane rotation, colour gains and the
    occlusion patch (x, y, w, h, noise (h, w, 3)) or None."""
    __slots__ = ("center", "scale", "rot", "gain", "patch")

    def __init__(self, center, scale, rot=0.0, gain=None, patch=None):
        self.center, self.scale, self.rot, self.gain, self.patch = center, scale, rot, gain, patch


def draw_occlusion_patch(center, scale, width, height, prob):
    """The random-occlusion draw of one view (upstream ``RandomOcclusion``, transform.py:21-66, applied to the jittered box at
    :111-116): consumes one ``np.random.rand()`` and, when the coin falls at or below ``prob``, four ``random.random()``
    values -- relative area (up to a fifth of the box), aspect ratio (0.5 .. 2), left edge, top edge -- and, if the patch lies
    inside the (width, height) image, ``np.random.rand(h, w, 3)``.  -> (x, y, w, h, noise * 255) or None.  Every expression
    keeps upstream's operand order: the fixture pins the patched pixels through their CRC."""
    if np.random.rand() > prob:
        return None
    left, top = center[0] - scale * 0.5, center[1] - scale * 0.5
    box_w, box_h = (left + scale) - left, (top + scale) - top           # (xmax - xmin as upstream forms it)
    rel_area, rel_aspect, rel_x, rel_y = random.random(), random.random(), random.random(), random.random()
    area = (rel_area * 0.2) * box_w * box_h
    aspect = rel_aspect * (2.0 - 0.5) + 0.5
    ph, pw = math.sqrt(area * aspect), math.sqrt(area / aspect)
    x = rel_x * (box_w - pw - 1) + left
    y = rel_y * (box_h - ph - 1) + top
    if not (x >= 0 and y >= 0 and x + pw < width and y + ph < height):
        return None
    x, y, pw, ph = int(x), int(y), int(pw), int(ph)
    return x, y, pw, ph, np.random.rand(ph, pw, 3) * 255


@TRANSFORM.register_module()
class SimpleTransform3DMultiView:
    """Counterpart of upstream's ``SimpleTransform2D`` + ``SimpleTransform3DMultiView`` (transform.py:70-196,240-281): same
    config keys, same result keys and dtypes, same consumption of ``np.random`` / ``random`` -- but organised per *frame*:

      :meth:`draw`          the random numbers of one view (the only sequential part)
      :meth:`frame_labels`  the label arithmetic of all views of a frame in one vectorised pass (:func:`crop_geometry`)
      :meth:`images`        the pixels of any number of views in ONE ``poem_warp_affine`` launch

    ``__call__(image, label, no_rot=False)`` is the per-view form with the reference's signature.  The random-occlusion patch
    (upstream default when the TRANSFORM node carries no OCCLUSION key; the released configs switch it off) is drawn with the
    view's other random numbers and written into a copy of the raw image before the upload.  Heat-map / mask targets belong to
    the training losses and are not built."""

    def __init__(self, cfg):
        self._output_size, self._train, self._aug = cfg.DATA_PRESET.IMAGE_SIZE, cfg.IS_TRAIN, cfg.AUG
        on = bool(self._aug)
        self._jitter = {"center": cfg.get("CENTER_JIT", 0), "scale": cfg.get("SCALE_JIT", 0.04 if on else 0),
                        "rot_deg": cfg.get("ROT_JIT", 10 if on else 0), "color": cfg.get("COLOR_JIT", 0.3 if on else 0)}
        self._rot_prob = cfg.get("ROT_PROB", 1.0 if on else 0)
        # upstream's defaults when the keys are absent: on with probability 0.1 under AUG (transform.py:83-84); the released
        # configs say OCCLUSION: False.  The patch is host-side work on the raw uint8 image, before the upload.
        self._occlusion_prob = cfg.get("OCCLUSION_PROB", 0.1 if on else 0) if cfg.get("OCCLUSION", on) else None
        if cfg.DATA_PRESET.get("WITH_HEATMAP", False) or cfg.DATA_PRESET.get("WITH_MASK", False):
            raise NotImplementedError("heat-map / mask targets are training-side and outside the built path")
        self.device = cfg.get("DEVICE", "cuda:0")

    # -- random numbers ---------------------------------------------------------------------------------------------------
    def draw(self, label, no_rot=False, image_shape=None):
        """Consumes the generators exactly as one upstream ``__call__`` does: four normal deviates from ``np.random``
        (centre x/y, scale, angle), one uniform unless ``no_rot`` (the master view keeps its orientation), the occlusion
        draw when OCCLUSION is on (:func:`draw_occlusion_patch`; needs the raw image's (H, W)) and -- from the ``random``
        module, an independent stream -- three colour gains."""
        if not self._aug:
            return ViewDraw(label["bbox_center"], label["bbox_scale"])
        j = self._jitter
        z = np.random.standard_normal(4)
        center = label["bbox_center"] + (0 + j["center"] * z[:2]) * label["bbox_scale"]
        scale = label["bbox_scale"] * (1 + j["scale"] * z[2])
        turn = (not no_rot) and np.random.rand() <= self._rot_prob
        rot = np.deg2rad(0 + j["rot_deg"] * z[3]) if turn else 0.0
        patch = None
        if self._occlusion_prob is not None:
            if image_shape is None:
                raise ValueError("OCCLUSION is on: draw() needs image_shape=(H, W) of the raw view")
            patch = draw_occlusion_patch(center, scale, image_shape[1], image_shape[0], self._occlusion_prob)
        lo, hi = 1 - j["color"], 1 + j["color"]
        return ViewDraw(center, scale, rot, [random.uniform(lo, hi) for _ in range(3)], patch)

    # -- labels -----------------------------------------------------------------------------------------------------------
    def frame_labels(self, images, labels, draws):
        """images / labels / draws: per kept view.  -> list of per-view result dicts (no pixels yet: ``raw_image``,
        ``color_gain`` are the inputs of :meth:`images`)."""
        V = len(labels)
        K = [lab["cam_intr"] for lab in labels]
        R, A, P = crop_geometry([d.center for d in draws], [d.scale for d in draws], [d.rot for d in draws],
                                [[k[0, 2], k[1, 2]] for k in K], self._output_size)
        W, H = self._output_size[0], self._output_size[1]
        out = []
        for v in range(V):
            lab, d = labels[v], draws[v]
            j2 = np.asarray(lab["joints_2d"])
            a = A[v].astype(np.float64)
            uv = np.stack([(a[0, 0] * j2[:, 0] + a[0, 1] * j2[:, 1]) + a[0, 2], (a[1, 0] * j2[:, 0] + a[1, 1] * j2[:, 1]) + a[1, 2]],
                          axis=1).astype(np.float32)
            if not self._train:
                vis = np.ones(NUM_JOINTS, dtype=np.float32)
            else:
                inside = (uv[:, 0] >= 0) & (uv[:, 0] < W) & (uv[:, 1] >= 0) & (uv[:, 1] < H)
                enough = lab["joints_vis"].sum() >= NUM_JOINTS * 0.3 and inside.sum() >= NUM_JOINTS * 0.3
                vis = inside.astype(np.float32) if enough else np.zeros(NUM_JOINTS, dtype=np.float32)
            p = P[v]
            k = np.asarray(K[v])
            dt = np.result_type(p.dtype, k.dtype)
            intr = np.stack([p[0, 0].astype(dt) * k[0].astype(dt) + p[0, 2].astype(dt) * k[2].astype(dt),
                             p[1, 1].astype(dt) * k[1].astype(dt) + p[1, 2].astype(dt) * k[2].astype(dt), k[2].astype(dt)])
            raw = images[v]
            if d.patch is not None:                     # (upstream writes into the caller's array; a copy is patched here)
                x, y, pw, ph, noise = d.patch
                raw = np.array(raw, copy=True)
                raw[y:y + ph, x:x + pw, :] = noise      # float -> uint8 truncation on assignment, as upstream
            out.append({"rot_rad": d.rot, "rot_mat3d": R[v], "affine": A[v], "target_bbox_center": d.center,
                        "target_bbox_scale": d.scale, "target_joints_2d": uv, "target_joints_vis": vis,
                        "image_path": lab["image_path"], "color_gain": d.gain, "raw_image": raw,
                        "affine_postrot": p, "extr_prerot": R[v], "target_cam_intr": intr,
                        "target_joints_3d": _rows_times(R[v], lab["joints_3d"]),
                        "target_verts_3d": _rows_times(R[v], lab["verts_3d"]),
                        "target_joints_3d_no_rot": lab["joints_3d"], "target_verts_3d_no_rot": lab["verts_3d"]})
        return out

    def labels(self, image, label, **kwargs):
        """One view: draw, then the label arithmetic (pure numpy, runs anywhere)."""
        return self.frame_labels([image], [label], [self.draw(label, kwargs.get("no_rot", False), np.shape(image)[:2])])[0]

    # -- pixels -----------------------------------------------------------------------------------------------------------
    def images(self, results_list):
        """One launch for the views whose label results are given; fills ``image`` and drops the staging keys."""
        gains = [r["color_gain"] for r in results_list]
        use_gain = any(g is not None for g in gains)
        out = warp_views([r["raw_image"] for r in results_list], [r["affine"][:2, :] for r in results_list],
                         (int(self._output_size[0]), int(self._output_size[1])),
                         gains=[g if g is not None else [1.0, 1.0, 1.0] for g in gains] if use_gain else None,
                         device=self.device)
        for i, r in enumerate(results_list):
            r["image"] = out[i]
            del r["raw_image"], r["color_gain"]
        return out

    def __call__(self, image, label, **kwargs):
        r = self.labels(image, label, **kwargs)
        self.images([r])
        return r
