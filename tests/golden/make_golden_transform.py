"""Golden vectors for SURVEY 8f row N4 from the upstream reference (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_transform.py

Runs the reference's own ``SimpleTransform3DMultiView`` (lib/utils/transform.py), ``MultiviewWebDataset.process_data_item``
(lib/data_wds/multiview_wds.py) and ``collation_random_n_views`` (lib/utils/collation.py) on the seeded synthetic records of
``oracle/transform_oracle.synthetic_frame`` and records every label-side output, plus the (matrix, size) arguments the
reference hands to ``cv2.warpAffine``.  OpenCV, torchvision and webdataset are absent from this image: ``cv2.warpAffine``
is replaced by a recorder that returns a black image (so NO pixel value in this fixture comes from the reference -- the
warp itself is "parity unpinned", see oracle/transform_oracle.py), ``tvF.to_tensor`` / ``tvF.normalize`` by their one-line
definitions, ``webdataset`` / ``braceexpand`` by mocks (only the record-processing method is called)."""
import os
import random
import sys
import zlib
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, HERE, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
sys.dont_write_bytecode = True

import ref_harness as rh  # noqa: E402
import transform_oracle as to  # noqa: E402

CASES = {
    # name: (dataset name in URLS, frames [(seed, n_cams)], random views, view range, aug, flip, label dtype, rng seed)
    "eval": ("DexYCB", [(1, 4), (2, 3)], False, None, False, False, np.float32, 0),
    "eval_random_views": ("Interhand", [(3, 8), (4, 6)], True, [2, 5], False, False, np.float32, 7),
    "train_aug": ("DexYCB", [(5, 4), (6, 2)], True, [1, 8], True, False, np.float64, 11),
    "flip": ("Oakink", [(7, 3)], False, None, False, True, np.float32, 13),
    # round 3: the reference's DEFAULT augmentation keys -- OCCLUSION absent from the TRANSFORM node means on, probability 0.1
    # (lib/utils/transform.py:83-84); a high probability here so that patches are actually drawn ("occl")
    "train_occlusion": ("DexYCB", [(8, 4), (9, 3)], True, [2, 6], "occl", False, np.float64, 17),
}
AUG = {"AUG": True, "CENTER_JIT": 0.05, "SCALE_JIT": 0.06, "ROT_JIT": 5, "COLOR_JIT": 0.3, "ROT_PROB": 0.5,
       "OCCLUSION": False, "OCCLUSION_PROB": 0.2}                      # config/release/train_medium.yaml:31-40
KEYS = ("affine", "affine_postrot", "rot_mat3d", "extr_prerot", "target_cam_intr", "target_cam_extr", "target_joints_2d",
        "target_joints_vis", "target_joints_3d", "target_joints_3d_no_rot", "target_bbox_center", "target_bbox_scale",
        "rot_rad", "mano_pose", "cam_extr", "idx", "master_joints_3d")


def main():
    sys.modules.setdefault("braceexpand", MagicMock())
    CN, _ = rh.setup()
    import lib.utils.transform as T
    import lib.data_wds.multiview_wds as MW
    from lib.utils.collation import collation_random_n_views

    calls = []

    def fake_warp(img, M, dsize, **kw):
        calls.append((np.array(M, dtype=np.float64), tuple(int(v) for v in dsize), zlib.crc32(np.ascontiguousarray(img).tobytes())))
        return np.zeros((int(dsize[1]), int(dsize[0]), 3), dtype=np.uint8)

    T.cv2 = MagicMock()
    T.cv2.warpAffine = fake_warp
    MW.cv2 = T.cv2
    T.tvF = MagicMock()
    T.tvF.to_tensor = lambda im: torch.from_numpy(np.ascontiguousarray(im.transpose(2, 0, 1))).float().div(255)
    T.tvF.normalize = lambda t, mean, std: (t - torch.tensor(mean)[:, None, None]) / torch.tensor(std)[:, None, None]

    rec = {}
    for name, (ds, frames, rnd, vr, aug, flip, dt, seed) in CASES.items():
        cfg = CN({"URLS": f"data/dataset_tars/{ds}_mv/{ds}_mv_test-{{000000..000003}}.tar", "DATA_SPLIT": "test",
                  "RANDOM_N_VIEWS": rnd, "VIEW_RANGE": vr,
                  "TRANSFORM": dict({"TYPE": "SimpleTransform3DMultiView"},
                                    **(dict(AUG, OCCLUSION=True, OCCLUSION_PROB=0.7) if aug == "occl" else AUG if aug else {"AUG": False})),
                  "DATA_PRESET": {"IMAGE_SIZE": [256, 256], "CENTER_IDX": 9}})
        dset = MW.MultiviewWebDataset(cfg, data_preset=cfg.DATA_PRESET, is_train=bool(aug))
        random.seed(seed)
        np.random.seed(seed)
        outs = []
        for fi, (fseed, ncam) in enumerate(frames):
            item = to.synthetic_frame(fseed, n_cams=ncam, dtype=dt)
            if flip:
                item["label.pyd"]["request_flip"] = True
            del calls[:]
            out = dset.process_data_item(item)
            outs.append(out)
            for k in KEYS:
                rec[f"{name}.{fi}.{k}"] = np.asarray(out[k])
            rec[f"{name}.{fi}.target_verts_3d_s16"] = np.asarray(out["target_verts_3d"])[:, ::16]
            rec[f"{name}.{fi}.warp_M"] = np.stack([c[0] for c in calls])
            rec[f"{name}.{fi}.warp_size"] = np.asarray([c[1] for c in calls])
            rec[f"{name}.{fi}.warp_src_crc"] = np.asarray([c[2] for c in calls], dtype=np.int64)   # the pixels handed to the warp (occlusion patches included)
            rec[f"{name}.{fi}.master_serial"] = np.asarray(out["master_serial"])
            rec[f"{name}.{fi}.image_shape"] = np.asarray(out["image"].shape)
        col = collation_random_n_views(outs)
        rec[f"{name}.col.cam_view_num"] = np.asarray(col["cam_view_num"])
        rec[f"{name}.col.target_cam_extr"] = col["target_cam_extr"].numpy()
        rec[f"{name}.col.tensor_keys"] = np.asarray(sorted(k for k, v in col.items() if isinstance(v, torch.Tensor)))
        rec[f"{name}.col.list_keys"] = np.asarray(sorted(k for k, v in col.items() if isinstance(v, list)))
    np.savez_compressed(os.path.join(HERE, "transform.npz"), **rec)
    print("transform.npz:", len(rec), "arrays,", os.path.getsize(os.path.join(HERE, "transform.npz")), "bytes")
    os.chdir(ROOT)


if __name__ == "__main__":
    main()
