"""Multi-view record shards: the on-disk format the reference evaluates from and the per-frame
processing in front of the model.

Mirrors ``lib/data_wds/multiview_wds.py`` (``MultiviewWebDataset``: :27-147), ``lib/datasets/mix_dataset.py:79-93``
(``MixWebDataset``) and ``lib/utils/collation.py:7-25`` (``collation_random_n_views``).  Upstream builds on the
``webdataset`` package (absent here); the shard format itself is plain POSIX tar and is read directly:

  one record = consecutive tar members that share a key; member name = ``<key>.<field>`` where the key ends at the first
  dot of the base name; fields of a multi-view frame: ``image_<i>.jpg`` | ``image_<i>.png`` (one per camera) and
  ``label.pyd`` (a pickled dict of per-camera lists)                                   [multiview_wds.py:63-75]

Decoding follows webdataset's ``decode("rgb8")``: images through PIL -> RGB uint8 (H, W, 3); ``.pyd`` through pickle.
The per-view image work (mirror warp, crop / warp / normalise) runs on the GPU, one launch per frame
(``transform.SimpleTransform3DMultiView.images``) or one per batch (``defer_images=True`` + ``collation_random_n_views``).
"""
import io
import json
import os
import pickle
import random
import re
import tarfile

import numpy as np
import torch

from .config import CN
from .transform import build_transform, warp_views

INV_EXTR_DATASETS = ['Interhand', 'Arctic', 'Oakink', 'Oakink2']         # multiview_wds.py:14
IMAGE_EXTS = ("jpg", "jpeg", "png", "ppm", "pgm", "pbm", "pnm")


# ---- urls ---------------------------------------------------------------------------------------------------------------
def braceexpand(pattern):
    """``a-{000..012}.tar`` / ``{x,y}`` expansion (the subset of the braceexpand package shard lists use)."""
    m = re.search(r"\{([^{}]*)\}", pattern)
    if not m:
        return [pattern]
    head, body, tail = pattern[:m.start()], m.group(1), pattern[m.end():]
    rng = re.fullmatch(r"(-?\d+)\.\.(-?\d+)(?:\.\.(-?\d+))?", body)
    if rng:
        a, b, step = rng.group(1), rng.group(2), int(rng.group(3) or 1)
        width = max(len(a), len(b)) if (a.startswith("0") and len(a) > 1) or (b.startswith("0") and len(b) > 1) else 0
        lo, hi = int(a), int(b)
        seq = range(lo, hi + 1, abs(step)) if lo <= hi else range(lo, hi - 1, -abs(step))
        parts = [str(v).zfill(width) for v in seq]
    elif "," in body:
        parts = body.split(",")
    else:
        parts = ["{" + body + "}"]
        return [head + parts[0] + t for t in braceexpand(tail)]
    return [head + p + t for p in parts for t in braceexpand(tail)]


def expand_urls(urls):                                                     # multiview_wds.py:17-24
    if isinstance(urls, str):
        urls = [urls]
    return [u for url in urls for u in braceexpand(os.path.expanduser(os.path.expandvars(url)))]


def split_by_node(urls, rank=None, world=None):
    """webdataset.split_by_node: rank r of w reads shards r, r+w, ... (one process per GPU; no data-path collective)."""
    if rank is None:
        rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    return list(urls)[rank::world] if world > 1 else list(urls)


# ---- records --------------------------------------------------------------------------------------------------------------
_KEY = re.compile(r"^((?:.*/|)[^.]+)[.]([^/]*)$")


def tar_records(url):
    """Raw records of one shard: dicts {"__key__", "__url__", <field>: bytes}, in file order."""
    cur = None
    with tarfile.open(url, "r|*") as tf:
        for member in tf:
            if not member.isreg():
                continue
            m = _KEY.match(member.name)
            if m is None:
                continue
            key, field = m.group(1), m.group(2).lower()
            if cur is None or cur["__key__"] != key:
                if cur is not None:
                    yield cur
                cur = {"__key__": key, "__url__": url}
            if field in cur:
                raise ValueError(f"{url}: duplicate field {field} in record {key}")
            cur[field] = tf.extractfile(member).read()
    if cur is not None:
        yield cur


def decode_rgb8(data):
    from PIL import Image
    with Image.open(io.BytesIO(data)) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def decode_record(rec):
    """webdataset ``decode("rgb8")``: images -> uint8 (H, W, 3); pyd / json / txt / cls / npy by extension."""
    out = {}
    for k, v in rec.items():
        if k.startswith("__"):
            out[k] = v
            continue
        ext = k.rsplit(".", 1)[-1]
        if ext in IMAGE_EXTS:
            out[k] = decode_rgb8(v)
        elif ext in ("pyd", "pickle"):
            out[k] = pickle.loads(v)
        elif ext in ("json", "jsn"):
            out[k] = json.loads(v)
        elif ext in ("txt", "text"):
            out[k] = v.decode("utf-8")
        elif ext in ("cls", "cls2", "class", "count", "index", "inx", "id"):
            out[k] = int(v)
        elif ext == "npy":
            out[k] = np.load(io.BytesIO(v), allow_pickle=False)
        else:
            out[k] = v
    return out


def write_shard(path, records, image_format="png", quality=95):
    """Inverse of the reader, for synthetic shards (tests, bench): records = decoded dicts as ``decode_record`` returns."""
    from PIL import Image
    with tarfile.open(path, "w") as tf:
        for rec in records:
            for k, v in rec.items():
                if k.startswith("__"):
                    continue
                ext = k.rsplit(".", 1)[-1]
                if ext in IMAGE_EXTS:
                    buf = io.BytesIO()
                    Image.fromarray(v).save(buf, format="PNG" if ext == "png" else "JPEG", **({} if ext == "png" else {"quality": quality}))
                    data = buf.getvalue()
                elif ext == "pyd":
                    data = pickle.dumps(v, protocol=4)
                else:
                    data = v if isinstance(v, bytes) else str(v).encode()
                info = tarfile.TarInfo(f"{rec['__key__']}.{k}")
                info.size = len(data)
                tf.addfile(info, io.BytesIO(data))


# ---- the dataset ------------------------------------------------------------------------------------------------------------
class MultiviewWebDataset:
    """Same constructor contract as upstream (``cfg.URLS / DATA_SPLIT / EPOCH_SIZE / RANDOM_N_VIEWS / VIEW_RANGE /
    TRANSFORM``, ``data_preset``, ``is_train``); iterating yields processed frames (``process_data_item``).

    ``defer_images=True`` leaves the pixels of a frame un-warped (``raw_image`` / ``affine`` / ``color_gain`` lists) so
    that ``collation_random_n_views`` can warp every view of the batch in one launch."""

    def __init__(self, cfg, data_preset=None, is_train=True, defer_images=False, rank=None, world=None):
        self.cfg, self.is_train, self.defer_images = cfg, is_train, defer_images
        self.urls, self.data_split = cfg.URLS, cfg.DATA_SPLIT
        self.epoch_size = cfg.get("EPOCH_SIZE", None)
        self.data_preset = cfg.DATA_PRESET if data_preset is None else data_preset
        self.mode = "train" if is_train else "val"
        # "<root>/<Name>_mv/<Name>_mv_<split>-{000000..N}.tar": the dataset name decides the extrinsics convention
        self.name = os.path.basename(cfg.URLS).split("_")[0]
        self.random_n_views, self.view_range = cfg.get("RANDOM_N_VIEWS", False), cfg.get("VIEW_RANGE", None)
        self.inv_extr = self.name in INV_EXTR_DATASETS
        if self.random_n_views and (self.view_range is None or self.view_range[0] < 1):
            raise AssertionError("RANDOM_N_VIEWS needs VIEW_RANGE = [lo >= 1, hi]")
        self.transform = build_transform(cfg=cfg.TRANSFORM, data_preset=self.data_preset, is_train=is_train)
        self.shards = split_by_node(expand_urls(self.urls), rank, world)

    def __iter__(self):
        shards = list(self.shards)
        if self.is_train:
            random.shuffle(shards)
        frames = (decode_record(r) for url in shards for r in tar_records(url))
        if self.is_train:
            frames = _buffer_shuffle(frames, 1000)                          # dataset.shuffle(1000), multiview_wds.py:51-52
        n = 0
        for item in frames:
            if self.epoch_size is not None and n >= self.epoch_size:
                return
            n += 1
            yield self.process_data_item(item)

    def get_dataset(self):
        return self

    # -- one record -> one processed frame (multiview_wds.py:62-145 upstream) --------------------------------------------
    def _choose_views(self, n_cams):
        """Camera indices kept for this frame, master first.  Random-view mode consumes ``random`` as upstream does:
        one shuffle, one Gaussian draw (mean 4, sigma 2) clamped to VIEW_RANGE and to the cameras there are."""
        order = list(range(n_cams))
        if not self.random_n_views:
            return order
        random.shuffle(order)
        want = int(round(random.gauss(4, 2)))
        lo, hi = self.view_range[0], self.view_range[1]
        return order[:min(max(lo, want), hi, n_cams)]

    def _mirrored(self, pixels, cam_intr, raw_size):
        """``request_flip`` records (left hands stored un-mirrored): reflect every kept view about the vertical line through
        its principal point, x -> 2 cx - x, at the raw resolution -- all views that share a raw size in one launch."""
        flips = [np.array([[-1, 0, 2 * k[0, 2]], [0, 1, 0]], dtype=np.float32) for k in cam_intr]
        out = [None] * len(pixels)
        for size in sorted({tuple(s) for s in raw_size}):
            group = [i for i, s in enumerate(raw_size) if tuple(s) == size]
            warped = warp_views([pixels[i] for i in group], [flips[i] for i in group], size, device=self.transform.device,
                                out="u8").cpu().numpy()
            for j, i in enumerate(group):
                out[i] = warped[j]
        return out

    @staticmethod
    def remaster_extrinsics(cam_extr, prerot):
        """cam_extr (V,4,4): camera -> old-master transforms of the kept views, view 0 = the new master; prerot (V,3,3): the
        in-plane rotation applied to each view's image.  -> (V,4,4) fp32 camera -> new-master transforms of the *rotated*
        cameras:  inv( [R_v 0; 0 1] . inv( inv(T_0) . T_v ) ), batched over the views."""
        T = np.stack([np.asarray(t) for t in cam_extr])
        rel = np.linalg.inv(T[0]) @ T
        pre = np.zeros((len(T), 4, 4))
        pre[:, :3, :3] = prerot
        pre[:, 3, 3] = 1
        return np.linalg.inv(pre @ np.linalg.inv(rel)).astype(np.float32)

    def process_data_item(self, item):
        cams = {}                                                            # camera index -> decoded pixels
        for name, value in item.items():
            m = re.fullmatch(r"image_(\d+)\.(\w+)", name)
            if m:
                cams[int(m.group(1))] = value
        n_cams, labels = len(cams), item["label.pyd"]
        has_fit = "mano_pose" in labels
        # MANO fits are stored with trailing extras (keep 16 x 3); records without fits (Oakink dumps) get zeros
        labels["mano_pose"] = [np.asarray(labels["mano_pose"][c]).reshape(-1)[:48].reshape(16, 3) if has_fit else np.zeros((16, 3))
                               for c in range(n_cams)]
        if not has_fit:
            labels["mano_shape"] = [np.zeros(10)] * n_cams
        if self.inv_extr:                                                    # these datasets store master -> camera
            labels["cam_extr"] = list(np.linalg.inv(np.stack(labels["cam_extr"][:n_cams])))
        keep = self._choose_views(n_cams)
        master = keep[0]
        fields = [k for k in labels if k != "request_flip"]
        view_labels = [{k: labels[k][c] for k in fields} for c in keep]
        pixels = [cams[c] for c in keep]
        if labels.get("request_flip", False):
            pixels = self._mirrored(pixels, [lab["cam_intr"] for lab in view_labels], [lab["raw_size"] for lab in view_labels])
        draws = [self.transform.draw(lab, no_rot=(c == master), image_shape=np.shape(px)[:2]) for lab, c, px in zip(view_labels, keep, pixels)]
        views = self.transform.frame_labels(pixels, view_labels, draws)
        extr = self.remaster_extrinsics([lab["cam_extr"] for lab in view_labels], [v["extr_prerot"] for v in views])
        for v, lab, e in zip(views, view_labels, extr):
            v["target_cam_extr"] = e
            v.update(lab)
        if not self.defer_images:
            self.transform.images(views)
        frame = {k: _column([v[k] for v in views], keep_list=k in ("raw_image", "color_gain")) for k in views[0]}
        frame.update(master_id=0, master_serial=labels["cam_serial"][master], master_joints_3d=labels["joints_3d"][master],
                     master_verts_3d=labels["verts_3d"][master], __key__=item["__key__"])
        return frame


def _column(values, keep_list=False):
    """Per-view values of one field -> the frame's entry.  As upstream (`np.stack` when the first value is an int, a float
    or an array): numbers and arrays gain a leading view axis, everything else (paths, serials, size tuples, numpy scalars
    that are not Python floats) stays a per-view list; device images are stacked on the device."""
    first = values[0]
    if keep_list:
        return values
    if isinstance(first, torch.Tensor):
        return torch.stack(values)
    if isinstance(first, (int, float, np.ndarray)):
        return np.stack(values)
    return values


def _buffer_shuffle(it, size):
    buf = []
    for x in it:
        if len(buf) < size:
            buf.append(x)
            continue
        i = random.randrange(size)
        buf[i], x = x, buf[i]
        yield x
    random.shuffle(buf)
    yield from buf


class MixWebDataset:
    """mix_dataset.py:79-93: draws each next frame from one of the member datasets with probability MIX_RATIO
    (webdataset.RandomMix: stops when the drawn source is exhausted), at most EPOCH_SIZE frames."""

    def __init__(self, cfg, dataset_list=None, max_len=None, data_preset=None, is_train=True, **kw):
        names = list(dataset_list if dataset_list is not None else cfg.DATASET_LIST)
        sub = [getattr(cfg, n) for n in names]
        self.datasets = [MultiviewWebDataset(c, data_preset, is_train, **kw) for c in sub]
        r = np.array([c.MIX_RATIO for c in sub], dtype=np.float64)
        self.ratios = r / r.sum()
        self.length = cfg.EPOCH_SIZE

    def __len__(self):
        return self.length

    def __iter__(self):
        sources = [iter(d) for d in self.datasets]
        cum = np.cumsum(self.ratios)
        for _ in range(self.length):
            i = min(int(np.searchsorted(cum, random.random())), len(sources) - 1)
            try:
                yield next(sources[i])
            except StopIteration:
                return


def collation_random_n_views(batch, transform=None):
    """collation.py:7-25.  numpy fields of the frames are concatenated along the view axis and become fp32 tensors, the
    rest become per-frame lists; ``cam_view_num`` (B,) is added.  Frames produced with ``defer_images=True`` carry raw
    pixels: pass the dataset's ``transform`` and every view of the batch is warped by one launch here."""
    if not isinstance(batch, list):
        batch = [batch]
    out = {}
    views = [b["target_joints_3d"].shape[0] for b in batch]
    if "raw_image" in batch[0]:
        if transform is None:
            raise ValueError("frames with deferred images need the dataset's transform")
        flat = [{"raw_image": im, "affine": a, "color_gain": g} for b in batch
                for im, a, g in zip(b["raw_image"], b["affine"], b["color_gain"])]
        out["image"] = transform.images(flat)
    for k in batch[0]:
        if k in ("raw_image", "color_gain"):
            continue
        v0 = batch[0][k]
        if isinstance(v0, torch.Tensor):
            out[k] = torch.cat([b[k] for b in batch], dim=0)
        elif isinstance(v0, np.ndarray) and not isinstance(v0[0], str):
            out[k] = torch.Tensor(np.concatenate([b[k] for b in batch], axis=0))
        else:
            out[k] = [b[k] for b in batch]
    out["cam_view_num"] = np.array(views)
    return out


def dataset_cfg(urls, view_range=None, image_size=(256, 256), aug=False, epoch_size=None, device="cuda:0", **tf):
    """A dataset node of the shape the release YAMLs hold (config/release/*.yaml DATASET.TEST.<name>)."""
    return CN({"URLS": urls, "DATA_SPLIT": "test", "EPOCH_SIZE": epoch_size, "RANDOM_N_VIEWS": view_range is not None,
               "VIEW_RANGE": list(view_range) if view_range is not None else None,
               "TRANSFORM": {"TYPE": "SimpleTransform3DMultiView", "AUG": aug, "DEVICE": device, **tf},
               "DATA_PRESET": {"IMAGE_SIZE": list(image_size), "CENTER_IDX": 9}})
