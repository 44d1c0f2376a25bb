"""Input-side transform of the multi-view pipeline: ``SimpleTransform3DMultiView`` with the image
work (crop / warp / colour jitter / to-tensor / normalise) on the MI355X and the label arithmetic on the host.

Mirrors ``lib/utils/transform.py`` upstream: ``SimpleTransform2D.__call__`` (:105-196), ``SimpleTransform3DMultiView``
(:240-281), the affine helpers (:618-705) and ``RandomOcclusion`` (:21-67); same config keys, same result keys, the same
draws from ``np.random`` / ``random`` in the same order (so a seeded upstream run and a seeded run here pick the same
augmentation).  The image chain of the reference is ``cv2.warpAffine`` -> colour jitter -> ``tvF.to_tensor`` ->
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


# ---- affine helpers (transform.py:618-705) --------------------------------------------------------------------------
def _construct_rotation_matrix(rot, size=3):
    m = np.eye(size, dtype=np.float32)
    if rot != 0:
        sn, cs = np.sin(rot), np.cos(rot)
        m[0, :2] = [cs, -sn]
        m[1, :2] = [sn, cs]
    return m


def _get_affine_trans_no_rot(center, scale, res):
    a = np.zeros((3, 3))
    ratio = float(res[0]) / float(res[1])
    a[0, 0] = float(res[0]) / scale
    a[1, 1] = float(res[1]) / scale * ratio
    a[0, 2] = res[0] * (-float(center[0]) / scale + 0.5)
    a[1, 2] = res[1] * (-float(center[1]) / scale * ratio + 0.5)
    a[2, 2] = 1
    return a


def _affine_transform(center, scale, out_res, rot=0):
    rotmat = _construct_rotation_matrix(rot=rot, size=3)
    origin_rot_center = (rotmat.dot(np.concatenate([center, np.ones(1)])))[:2]
    return _get_affine_trans_no_rot(origin_rot_center, scale, out_res).dot(rotmat).astype(np.float32)


def _affine_transform_post_rot(center, scale, optical_center, out_res, rot=0):
    rotmat = _construct_rotation_matrix(rot=rot, size=3)
    t_mat = np.eye(3)
    t_mat[0, 2] = -optical_center[0]
    t_mat[1, 2] = -optical_center[1]
    t_inv = t_mat.copy()
    t_inv[:2, 2] *= -1
    c = t_inv.dot(rotmat).dot(t_mat).dot(np.concatenate([center, np.ones(1)]))
    return _get_affine_trans_no_rot(c[:2], scale, out_res).astype(np.float32)


def _transform_coords(pts, affine_trans, invert=False):
    if invert:
        affine_trans = np.linalg.inv(affine_trans)
    hom2d = np.concatenate([pts, np.ones([np.array(pts).shape[0], 1])], 1)
    return affine_trans.dot(hom2d.transpose()).transpose()[:, :2]


def center_scale_to_box(center, scale):               # transform.py:1083-1101
    w = h = scale * 1.0
    xmin = center[0] - w * 0.5
    ymin = center[1] - h * 0.5
    return [xmin, ymin, xmin + w, ymin + h]


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


# ---- RandomOcclusion (transform.py:21-67): modifies the raw image on the host, before the upload -------------------------
class RandomOcclusion:

    def __init__(self, occlusion_prob=0.5):
        self.occlusion_prob = occlusion_prob

    def __call__(self, results):
        if np.random.rand() > self.occlusion_prob:
            return results
        xmin, ymin, xmax, ymax = results["bbox"]
        imgwidth, imgheight, img = results["width"], results["height"], results["image"]
        synth_area = (random.random() * 0.2) * (xmax - xmin) * (ymax - ymin)
        synth_ratio = random.random() * (1 / 0.5 - 0.5) + 0.5
        synth_h = math.sqrt(synth_area * synth_ratio)
        synth_w = math.sqrt(synth_area / synth_ratio)
        synth_xmin = random.random() * ((xmax - xmin) - synth_w - 1) + xmin
        synth_ymin = random.random() * ((ymax - ymin) - synth_h - 1) + ymin
        if synth_xmin >= 0 and synth_ymin >= 0 and synth_xmin + synth_w < imgwidth and synth_ymin + synth_h < imgheight:
            x0, y0, w, h = int(synth_xmin), int(synth_ymin), int(synth_w), int(synth_h)
            img[y0:y0 + h, x0:x0 + w, :] = (np.random.rand(h, w, 3) * 255)
        results["image"] = img
        return results


# ---- the transform ------------------------------------------------------------------------------------------------------
@TRANSFORM.register_module()
class SimpleTransform3DMultiView:
    """``SimpleTransform2D`` + ``SimpleTransform3DMultiView`` (transform.py:70-196,240-281).

    ``__call__(image, label, no_rot=False)`` -> the reference's result dict (``image`` is a (3,H,W) fp32 device tensor).
    The batched form -- :meth:`labels` per view, then one :func:`warp_views` over all views of a frame or a batch -- is
    what ``MultiviewWebDataset`` uses.  Heat-map / mask targets (``WITH_HEATMAP`` / ``WITH_MASK``) belong to the training
    losses and are not built."""

    def __init__(self, cfg):
        self._output_size = cfg.DATA_PRESET.IMAGE_SIZE
        self._train = cfg.IS_TRAIN
        self._aug = cfg.AUG
        self._center_jit_factor = cfg.get("CENTER_JIT", 0)
        self._scale_jit_factor = cfg.get("SCALE_JIT", 0.04 if self._aug else 0)
        self._color_jit_factor = cfg.get("COLOR_JIT", 0.3 if self._aug else 0)
        self._rot_jit_factor = cfg.get("ROT_JIT", 10 if self._aug else 0)
        self._rot_prob = cfg.get("ROT_PROB", 1.0 if self._aug else 0)
        self._occlusion = cfg.get("OCCLUSION", True if self._aug else False)
        self._occlusion_prob = cfg.get("OCCLUSION_PROB", 0.1 if self._aug else 0)
        if cfg.DATA_PRESET.get("WITH_HEATMAP", False) or cfg.DATA_PRESET.get("WITH_MASK", False):
            raise NotImplementedError("heat-map / mask targets are training-side and outside the built path")
        self.device = cfg.get("DEVICE", "cuda:0")
        if self._occlusion:
            self.occlusion_op = RandomOcclusion(self._occlusion_prob)

    def labels(self, image, label, **kwargs):
        """Everything of the reference's ``__call__`` except the pixels: draws the augmentation, applies the occlusion to
        the raw image (host, in place, as upstream), returns the result dict without ``image`` plus ``color_gain``
        (3 gains or None) for the device stage.  Pure numpy."""
        if self._aug:
            cf, sf, rf = self._center_jit_factor, self._scale_jit_factor, self._rot_jit_factor
            c_factor = np.random.normal(loc=0, scale=cf, size=2)
            bbox_center = label["bbox_center"] + c_factor * label["bbox_scale"]
            s_factor = np.random.normal(loc=1, scale=sf)
            bbox_scale = label["bbox_scale"] * s_factor
            r_factor = np.random.normal(loc=0, scale=rf)
            no_rot = kwargs.get("no_rot", False)
            rot = np.deg2rad(r_factor) if (not no_rot and np.random.rand() <= self._rot_prob) else 0.0
            if self._occlusion:
                occ = {"bbox": center_scale_to_box(bbox_center, bbox_scale), "width": image.shape[1],
                       "height": image.shape[0], "image": image}
                image = self.occlusion_op(occ)["image"]
        else:
            bbox_scale, bbox_center, rot = label["bbox_scale"], label["bbox_center"], 0.0
        rot_mat3d = _construct_rotation_matrix(rot)
        affine = _affine_transform(center=bbox_center, scale=bbox_scale, out_res=self._output_size, rot=rot)
        target_joints_2d = _transform_coords(label["joints_2d"], affine).astype(np.float32)
        jv = label["joints_vis"]
        if not self._train:
            vis = np.full(NUM_JOINTS, 1.0, dtype=np.float32)
        elif jv.sum() < NUM_JOINTS * 0.3:
            vis = np.full(NUM_JOINTS, 0.0, dtype=np.float32)
        else:
            t = target_joints_2d
            vis = (((t[:, 0] >= 0) & (t[:, 0] < self._output_size[0])) &
                   ((t[:, 1] >= 0) & (t[:, 1] < self._output_size[1]))).astype(np.float32)
            if vis.sum() < NUM_JOINTS * 0.3:
                vis = np.full(NUM_JOINTS, 0.0, dtype=np.float32)
        gain = None
        if self._aug:                                   # the three draws follow the warp upstream; nothing between them
            c_high, c_low = 1 + self._color_jit_factor, 1 - self._color_jit_factor          # consumes these generators
            gain = [random.uniform(c_low, c_high) for _ in range(3)]
        results = {"rot_rad": rot, "rot_mat3d": rot_mat3d, "affine": affine, "target_bbox_center": bbox_center,
                   "target_bbox_scale": bbox_scale, "target_joints_2d": target_joints_2d, "target_joints_vis": vis,
                   "image_path": label["image_path"], "color_gain": gain, "raw_image": image}
        # SimpleTransform3DMultiView (:245-281)
        intr = label["cam_intr"]
        cc = np.array([intr[0, 2], intr[1, 2]])
        affine_postrot = _affine_transform_post_rot(center=bbox_center, scale=bbox_scale, optical_center=cc,
                                                    out_res=self._output_size, rot=rot)
        results["affine_postrot"] = affine_postrot
        results["extr_prerot"] = rot_mat3d
        results["target_cam_intr"] = affine_postrot.dot(label["cam_intr"])
        results["target_joints_3d"] = rot_mat3d.dot(label["joints_3d"].transpose(1, 0)).transpose()
        results["target_verts_3d"] = rot_mat3d.dot(label["verts_3d"].transpose(1, 0)).transpose()
        results["target_joints_3d_no_rot"] = label["joints_3d"]
        results["target_verts_3d_no_rot"] = label["verts_3d"]
        return results

    def images(self, results_list):
        """One launch for the views whose :meth:`labels` results are given; fills ``image`` and drops the staging keys."""
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
