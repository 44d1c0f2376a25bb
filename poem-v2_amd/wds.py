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
        self.cfg = cfg
        self.data_split = cfg.DATA_SPLIT
        self.epoch_size = cfg.get("EPOCH_SIZE", None)
        self.data_preset = data_preset if data_preset is not None else cfg.DATA_PRESET
        self.urls = cfg.URLS
        self.name = cfg.URLS.split("/")[-1].split("_")[0]
        self.inv_extr = self.name in INV_EXTR_DATASETS
        self.random_n_views = cfg.get("RANDOM_N_VIEWS", False)
        self.view_range = cfg.get("VIEW_RANGE", None)
        self.mode = "train" if is_train else "val"
        self.is_train = is_train
        self.defer_images = defer_images
        self.transform = build_transform(cfg=cfg.TRANSFORM, data_preset=self.data_preset, is_train=is_train)
        if self.random_n_views:
            assert self.view_range is not None and self.view_range[0] >= 1
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

    def process_data_item(self, item):                                       # multiview_wds.py:62-145
        n_view_imgs = {k: v for k, v in item.items() if k.startswith("image")}
        img_type = "jpg"
        for k in n_view_imgs:
            img_type = "png" if "png" in k else "jpg"
        n_cams = len(n_view_imgs)
        key = item["__key__"]
        labels = item["label.pyd"]
        if "mano_pose" in labels:
            labels["mano_pose"] = [labels["mano_pose"][i].reshape(-1)[:48].reshape(16, 3) for i in range(n_cams)]
        else:
            labels["mano_pose"] = [np.zeros((16, 3)) for _ in range(n_cams)]
            labels["mano_shape"] = [np.zeros(10) for _ in range(n_cams)]
        if self.inv_extr:
            labels["cam_extr"] = [np.linalg.inv(labels["cam_extr"][i]) for i in range(n_cams)]
        indices = list(range(n_cams))
        if self.random_n_views:
            random.shuffle(indices)
            n = int(round(random.gauss(4, 2)))
            n = min(max(self.view_range[0], n), self.view_range[1])
            indices_keep = indices[:min(n, n_cams)]
        else:
            indices_keep = indices
        new_master_id = indices_keep[0]
        T_master_2_new_master = labels["cam_extr"][new_master_id]
        imgs = [n_view_imgs[f"image_{ind}.{img_type}"] for ind in indices_keep]
        if labels.get("request_flip", False):                                # :112-118, all kept views in one launch
            flips = [np.array([[-1, 0, 2 * labels["cam_intr"][ind][0, 2]], [0, 1, 0]], dtype=np.float32) for ind in indices_keep]
            sizes = {tuple(labels["raw_size"][ind]) for ind in indices_keep}
            if len(sizes) == 1:
                imgs = list(warp_views(imgs, flips, sizes.pop(), device=self.transform.device, out="u8").cpu().numpy())
            else:
                imgs = [warp_views([im], [M], labels["raw_size"][ind], device=self.transform.device, out="u8")[0].cpu().numpy()
                        for im, M, ind in zip(imgs, flips, indices_keep)]
        per_view = []
        for img, ind in zip(imgs, indices_keep):
            lab = {k: v[ind] for k, v in labels.items() if k not in ["request_flip"]}
            tgt = self.transform.labels(img, lab, no_rot=ind == new_master_id)
            T_new_master_2_cam = np.linalg.inv(T_master_2_new_master) @ lab["cam_extr"]
            pre = np.concatenate([tgt["extr_prerot"], np.zeros((3, 1))], axis=1)
            pre = np.concatenate([pre, np.array([[0, 0, 0, 1]])], axis=0)
            tgt["target_cam_extr"] = np.linalg.inv(pre @ np.linalg.inv(T_new_master_2_cam)).astype(np.float32)
            tgt.update(lab)
            per_view.append(tgt)
        if not self.defer_images:
            self.transform.images(per_view)
        res = {}
        for tgt in per_view:
            for k, v in tgt.items():
                res.setdefault(k, []).append(v)
        for q in res:
            if q in ("raw_image", "color_gain"):
                continue
            if isinstance(res[q][0], torch.Tensor):
                res[q] = torch.stack(res[q])                                 # device images stay on the device
            elif isinstance(res[q][0], (int, float, np.ndarray)):
                res[q] = np.stack(res[q])
        res["master_id"] = 0
        res["master_serial"] = labels["cam_serial"][new_master_id]
        res["master_joints_3d"] = labels["joints_3d"][new_master_id]
        res["master_verts_3d"] = labels["verts_3d"][new_master_id]
        res["__key__"] = key
        return res


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
