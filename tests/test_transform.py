"""Input side: shard records -> per-view transform -> batch.

CPU: the label arithmetic of the oracle AND of the product against the reference's own outputs (tests/golden/transform.npz),
the tar record format, the oracle's warp against the defining properties of OpenCV's fixed-point scheme.
GPU: ``poem_warp_affine`` (through the C ABI) bit-exact against the oracle; the whole pipeline shard -> frames -> batch."""
import os
import random

import numpy as np
import pytest
import torch

import transform_oracle as to
from util import GOLDEN

DEV = "cuda:0"
AUG = {"AUG": True, "CENTER_JIT": 0.05, "SCALE_JIT": 0.06, "ROT_JIT": 5, "COLOR_JIT": 0.3, "ROT_PROB": 0.5,
       "OCCLUSION": False, "OCCLUSION_PROB": 0.2}
CASES = {   # mirrors tests/golden/make_golden_transform.py::CASES
    "eval": ("DexYCB", [(1, 4), (2, 3)], False, None, False, False, np.float32, 0),
    "eval_random_views": ("Interhand", [(3, 8), (4, 6)], True, [2, 5], False, False, np.float32, 7),
    "train_aug": ("DexYCB", [(5, 4), (6, 2)], True, [1, 8], True, False, np.float64, 11),
    "flip": ("Oakink", [(7, 3)], False, None, False, True, np.float32, 13),
    # upstream's default augmentation keys include the random occlusion patch (OCCLUSION absent = on, lib/utils/transform.py:83-84)
    "train_occlusion": ("DexYCB", [(8, 4), (9, 3)], True, [2, 6], "occl", False, np.float64, 17),
}
KEYS = ("affine", "affine_postrot", "rot_mat3d", "extr_prerot", "target_cam_intr", "target_cam_extr", "target_joints_2d",
        "target_joints_vis", "target_joints_3d", "target_joints_3d_no_rot", "target_bbox_center", "target_bbox_scale",
        "rot_rad", "mano_pose", "cam_extr", "idx", "master_joints_3d")


def _golden():
    return np.load(os.path.join(GOLDEN, "transform.npz"))


def _frames(case):
    ds, frames, rnd, vr, aug, flip, dt, seed = CASES[case]
    for fseed, ncam in frames:
        item = to.synthetic_frame(fseed, n_cams=ncam, dtype=dt)
        if flip:
            item["label.pyd"]["request_flip"] = True
        yield item


def _oracle_aug(aug):
    if not aug:
        return None
    d = dict(center_jit=0.05, scale_jit=0.06, rot_jit=5, rot_prob=0.5, color_jit=0.3)
    if aug == "occl":
        d["occlusion_prob"] = 0.7
    return d


def _crc(img):
    import zlib
    return zlib.crc32(np.ascontiguousarray(img).tobytes())


def _same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    assert np.array_equal(a, b), (what, float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max()))


@pytest.mark.parametrize("case", list(CASES))
def test_oracle_labels_match_reference(case, monkeypatch):
    z = _golden()
    ds, frames, rnd, vr, aug, flip, dt, seed = CASES[case]
    random.seed(seed)
    np.random.seed(seed)
    tf = dict(is_train=bool(aug), aug=_oracle_aug(aug))
    warped_src, real_warp = [], to.warp_affine_u8
    monkeypatch.setattr(to, "warp_affine_u8", lambda img, M, size: (warped_src.append(_crc(img)), real_warp(img, M, size))[1])
    for fi, item in enumerate(_frames(case)):
        del warped_src[:]
        out = to.process_data_item(item, inv_extr=ds in ("Interhand", "Arctic", "Oakink", "Oakink2"), random_n_views=rnd,
                                   view_range=vr, **tf)
        if not flip:            # the pixels handed to the warp (occlusion patches included) are the reference's, byte for byte
            assert warped_src == z[f"{case}.{fi}.warp_src_crc"].tolist(), (case, fi)
        for k in KEYS:
            _same(out[k], z[f"{case}.{fi}.{k}"], f"{case}.{fi}.{k}")
        _same(np.asarray(out["target_verts_3d"])[:, ::16], z[f"{case}.{fi}.target_verts_3d_s16"], "verts")
        assert str(out["master_serial"]) == str(z[f"{case}.{fi}.master_serial"])
        assert tuple(out["image"].shape) == tuple(z[f"{case}.{fi}.image_shape"])


def _dataset(case, defer, device=DEV):
    import poem_v2_amd as pk
    ds, frames, rnd, vr, aug, flip, dt, seed = CASES[case]
    keys = {k: v for k, v in AUG.items() if k != "AUG"} if aug else {}
    if aug == "occl":
        keys.update(OCCLUSION=True, OCCLUSION_PROB=0.7)
    cfg = pk.wds.dataset_cfg(f"data/dataset_tars/{ds}_mv/{ds}_mv_test-{{000000..000003}}.tar", view_range=vr, device=device, **keys)
    cfg.TRANSFORM.AUG = bool(aug)
    return pk.MultiviewWebDataset(cfg, data_preset=cfg.DATA_PRESET, is_train=bool(aug), defer_images=defer)


@pytest.mark.parametrize("case", ["eval", "eval_random_views", "train_aug", "train_occlusion"])
def test_product_labels_match_reference(case):
    """The host side of the product (no GPU needed with deferred pixels) against the reference's outputs, and the
    matrices it will hand to the warp against the ones the reference handed to cv2.warpAffine."""
    import poem_v2_amd as pk
    z = _golden()
    ds, frames, rnd, vr, aug, flip, dt, seed = CASES[case]
    dset = _dataset(case, defer=True)
    assert len(dset.shards) == 4 and dset.shards[3].endswith("-000003.tar") and dset.inv_extr == (ds == "Interhand")
    random.seed(seed)
    np.random.seed(seed)
    outs = []
    for fi, item in enumerate(_frames(case)):
        out = dset.process_data_item(item)
        outs.append(out)
        for k in KEYS:
            _same(out[k], z[f"{case}.{fi}.{k}"], f"{case}.{fi}.{k}")
        _same(np.asarray(out["target_verts_3d"])[:, ::16], z[f"{case}.{fi}.target_verts_3d_s16"], "verts")
        M = np.stack([np.asarray(a, np.float64)[:2] for a in out["affine"]])
        _same(M, z[f"{case}.{fi}.warp_M"], "warp matrices")
        assert len(out["raw_image"]) == len(M) and (out["color_gain"][0] is not None) == bool(aug)
        assert [_crc(im) for im in out["raw_image"]] == z[f"{case}.{fi}.warp_src_crc"].tolist()      # incl. the occlusion patches
    with pytest.raises(ValueError):
        pk.collation_random_n_views(outs)                       # deferred pixels need the transform
    for o in outs:                                              # label-only collation
        del o["raw_image"], o["color_gain"]
    col = pk.collation_random_n_views(outs)
    assert np.array_equal(col["cam_view_num"], z[f"{case}.col.cam_view_num"])
    _same(col["target_cam_extr"].numpy(), z[f"{case}.col.target_cam_extr"], "collated extrinsics")
    tk = sorted(k for k, v in col.items() if isinstance(v, torch.Tensor))
    lk = sorted(k for k, v in col.items() if isinstance(v, list))
    assert tk == [k for k in z[f"{case}.col.tensor_keys"] if k != "image"]
    assert lk == list(z[f"{case}.col.list_keys"])


def test_warp_oracle_defining_properties():
    """OpenCV is absent (parity unpinned): pin the restatement to what the fixed-point scheme *means*."""
    g = np.random.default_rng(0)
    img = g.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    I = np.array([[1, 0, 0], [0, 1, 0]], np.float32)
    assert np.array_equal(to.warp_affine_u8(img, I, (53, 37)), img)                               # identity
    sh = to.warp_affine_u8(img, np.array([[1, 0, 5], [0, 1, -3]], np.float32), (53, 37))         # integer shift
    assert np.array_equal(sh[:34, 5:], img[3:, :48]) and not sh[:, :5].any() and not sh[34:].any()
    fl = to.warp_affine_u8(img, np.array([[-1, 0, 52], [0, 1, 0]], np.float32), (53, 37))        # mirror (:112-118 upstream)
    assert np.array_equal(fl, img[:, ::-1])
    # general map: equals float64 bilinear interpolation at the 1/32-pixel source coordinates, rounded half up
    M = np.array([[1.37 * np.cos(0.3), -1.37 * np.sin(0.3), -7.3], [1.37 * np.sin(0.3), 1.37 * np.cos(0.3), 4.9]], np.float32)
    out = to.warp_affine_u8(img, M, (64, 48))
    sx, sy, ax, ay = to.fixed_point_coords(M, 64, 48)
    pad = np.zeros((37 + 4, 53 + 4, 3))
    pad[2:-2, 2:-2] = img
    ok = (sx >= -1) & (sx < 53) & (sy >= -1) & (sy < 37)
    sxc, syc = np.clip(sx, -2, 53) + 2, np.clip(sy, -2, 37) + 2
    fx, fy = (ax / 32.0)[..., None], (ay / 32.0)[..., None]
    ref = (pad[syc, sxc] * (1 - fx) * (1 - fy) + pad[syc, sxc + 1] * fx * (1 - fy) + pad[syc + 1, sxc] * (1 - fx) * fy
           + pad[syc + 1, sxc + 1] * fx * fy)
    ref = np.where(ok[..., None], np.floor(ref + 0.5), 0).astype(np.uint8)
    assert np.array_equal(out, ref)
    # the sub-pixel coordinates are the fp64 inverse map rounded to 1/32 pixel
    inv = np.linalg.inv(np.vstack([M.astype(np.float64), [0, 0, 1]]))
    xs, ys = np.meshgrid(np.arange(64.0), np.arange(48.0))
    assert np.abs((sx + ax / 32.0) - (inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2])).max() <= 1 / 64 + 2 / 1024
    assert np.abs((sy + ay / 32.0) - (inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2])).max() <= 1 / 64 + 2 / 1024


def test_to_tensor_normalize_matches_torch_on_every_byte():
    img = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, axis=2)
    t = torch.from_numpy(img.transpose(2, 0, 1).copy()).float().div(255)
    t = (t - torch.tensor([0.5, 0.5, 0.5])[:, None, None]) / torch.tensor([1.0, 1.0, 1.0])[:, None, None]
    assert np.array_equal(to.to_tensor_normalize(img), t.numpy())


def test_color_jitter_truncates_like_a_uint8_assignment():
    img = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, axis=2)
    out = to.color_jitter_u8(img, [0.7137, 1.2999, 1.0])
    assert out[..., 0].max() == int(255 * 0.7137) and out[..., 1].max() == 255 and np.array_equal(out[..., 2], img[..., 2])


def test_product_record_generator_equals_test_generator():
    """``scripts/eval_single.py --shards`` writes its shards with the product's own generator (poem_v2_amd.inputs); it is
    the same seeded record the test infrastructure uses, so fixtures and product shards describe the same frames."""
    import poem_v2_amd as pk

    def same(a, b):
        if isinstance(a, dict):
            return a.keys() == b.keys() and all(same(a[k], b[k]) for k in a)
        if isinstance(a, (list, tuple)):
            return len(a) == len(b) and all(same(x, y) for x, y in zip(a, b))
        if isinstance(a, np.ndarray):
            return a.dtype == b.dtype and np.array_equal(a, b)
        return type(a) is type(b) and a == b
    for kw in (dict(seed=3, n_cams=4), dict(seed=9, n_cams=2, raw=(96, 64), ext="jpg", dtype=np.float64)):
        assert same(pk.inputs.synthetic_frame(**kw), to.synthetic_frame(**kw))


def test_crop_geometry_is_the_composition_it_claims():
    """Independent of the fixtures: A maps the bbox square onto the window after rotating about the origin, P is a pure
    crop whose centre is the bbox centre rotated about the principal point."""
    import poem_v2_amd as pk
    g = np.random.default_rng(0)
    c, s, r, o = g.uniform(100, 400, (5, 2)), g.uniform(80, 200, 5), g.uniform(-0.3, 0.3, 5), g.uniform(200, 300, (5, 2))
    r[0] = 0.0
    R, A, P = pk.transform.crop_geometry(c, s, r, o, (256, 192))
    for v in range(5):
        Rm = np.array([[np.cos(r[v]), -np.sin(r[v])], [np.sin(r[v]), np.cos(r[v])]])
        assert np.allclose(R[v][:2, :2], Rm, atol=1e-7) and R[v][2, 2] == 1
        centre = A[v].astype(np.float64) @ np.array([c[v, 0], c[v, 1], 1.0])
        assert np.allclose(centre[:2], [128, 96], atol=1e-3)                      # bbox centre -> window centre
        assert np.allclose(A[v][:2, :2], (256 / s[v]) * Rm, rtol=1e-6, atol=1e-6)  # isotropic zoom (aspect folded in)
        pc = o[v] + Rm @ (c[v] - o[v])
        assert np.allclose(P[v].astype(np.float64) @ np.array([pc[0], pc[1], 1.0]), [128, 96, 1], atol=1e-3)
        assert P[v][0, 1] == 0 and P[v][1, 0] == 0


def test_shard_format_round_trip(tmp_path):
    import poem_v2_amd as pk
    recs = [to.synthetic_frame(s, n_cams=n, raw=(96, 64)) for s, n in ((1, 2), (2, 3))]
    jrec = to.synthetic_frame(3, n_cams=1, raw=(96, 64), ext="jpg")
    path = str(tmp_path / "Toy_mv_test-000000.tar")
    pk.wds.write_shard(path, recs + [jrec])
    raw = list(pk.wds.tar_records(path))
    assert [r["__key__"] for r in raw] == ["frame000001", "frame000002", "frame000003"]
    assert sorted(k for k in raw[1] if not k.startswith("__")) == ["image_0.png", "image_1.png", "image_2.png", "label.pyd"]
    dec = [pk.wds.decode_record(r) for r in raw]
    for d, r in zip(dec, recs):
        for k, v in r.items():
            if k.startswith("image"):
                assert d[k].dtype == np.uint8 and np.array_equal(d[k], v)            # PNG is lossless
        assert np.array_equal(d["label.pyd"]["cam_intr"][0], r["label.pyd"]["cam_intr"][0])
        assert d["label.pyd"]["cam_serial"] == r["label.pyd"]["cam_serial"]
    assert dec[2]["image_0.jpg"].shape == (64, 96, 3)                                 # JPEG: decoded through PIL, lossy
    assert np.abs(dec[2]["image_0.jpg"].astype(int) - jrec["image_0.jpg"].astype(int)).mean() < 25


def test_urls_and_node_split():
    import poem_v2_amd as pk
    u = pk.wds.expand_urls("data/x_mv/x_mv_val-{000000..000012}.tar")
    assert len(u) == 13 and u[0].endswith("-000000.tar") and u[12].endswith("-000012.tar")
    assert pk.wds.expand_urls(["a{1..3}", "b{x,y}c"]) == ["a1", "a2", "a3", "bxc", "byc"]
    assert pk.wds.expand_urls("plain.tar") == ["plain.tar"]
    assert pk.wds.split_by_node(u, 1, 4) == u[1::4] and pk.wds.split_by_node(u, 0, 1) == u
    parts = [pk.wds.split_by_node(u, r, 8) for r in range(8)]
    assert sorted(sum(parts, [])) == sorted(u)


def test_mix_dataset_draws_by_ratio(tmp_path):
    import poem_v2_amd as pk
    cfgs = {}
    for name, n in (("Aa", 30), ("Bb", 30)):
        p = str(tmp_path / f"{name}_mv_test-000000.tar")
        pk.wds.write_shard(p, [dict(to.synthetic_frame(100 + i, n_cams=1, raw=(32, 32)), __key__=f"{name}{i:03d}") for i in range(n)])
        c = pk.wds.dataset_cfg(p)
        c.MIX_RATIO = 3.0 if name == "Aa" else 1.0
        cfgs[name] = c
    mix = pk.MixWebDataset(pk.CN({"EPOCH_SIZE": 24, "DATASET_LIST": ["Aa", "Bb"], **cfgs}), is_train=False, defer_images=True)
    random.seed(0)
    keys = [f["__key__"] for f in mix]
    assert len(keys) == 24 and len(set(keys)) == 24
    na = sum(k.startswith("Aa") for k in keys)
    assert 12 <= na <= 23 and [k for k in keys if k.startswith("Aa")] == sorted(k for k in keys if k.startswith("Aa"))


# ---- GPU -----------------------------------------------------------------------------------------------------------------
def _maps(g, n, raw):
    out = []
    for i in range(n):
        ang, s = g.normal() * 0.4, float(np.exp(g.normal() * 0.5))
        A = np.array([[s * np.cos(ang), -s * np.sin(ang), g.normal() * 40 - 20], [s * np.sin(ang), s * np.cos(ang), g.normal() * 40 - 20]])
        out.append(A.astype(np.float32))
    return out


@pytest.mark.gpu
def test_warp_kernel_bit_exact_against_oracle():
    import poem_v2_amd as pk
    g = np.random.default_rng(5)
    sizes = [(64, 96), (480, 640), (37, 53), (1, 1), (200, 17), (300, 400)]
    imgs = [g.integers(0, 256, size=(h, w, 3), dtype=np.uint8) for h, w in sizes]
    Ms = _maps(g, len(imgs), None)
    Ms[0] = np.array([[1, 0, 0], [0, 1, 0]], np.float32)
    Ms[3] = np.array([[50, 0, 100], [0, 50, 100]], np.float32)                      # a single source pixel, mostly border
    gains = g.uniform(0.7, 1.3, size=(len(imgs), 3))
    for out_size in ((256, 256), (100, 60), (101, 7), (3, 5)):
        u8 = pk.transform.warp_views(imgs, Ms, out_size, device=DEV, out="u8").cpu().numpy()
        f32 = pk.transform.warp_views(imgs, Ms, out_size, device=DEV).cpu().numpy()
        ug = pk.transform.warp_views(imgs, Ms, out_size, gains=gains, device=DEV, out="u8").cpu().numpy()
        for i, (im, M) in enumerate(zip(imgs, Ms)):
            ref = to.warp_affine_u8(im, M, out_size)
            assert np.array_equal(u8[i], ref), (i, out_size)
            assert np.array_equal(f32[i], to.to_tensor_normalize(ref)), (i, out_size)
            assert np.array_equal(ug[i], to.color_jitter_u8(ref, gains[i])), (i, out_size)
    # every byte value through the fp32 tail
    ramp = np.arange(256, dtype=np.uint8).reshape(16, 16, 1).repeat(3, axis=2)
    t = pk.transform.warp_views([ramp], [np.eye(3, dtype=np.float32)], (16, 16), device=DEV).cpu().numpy()[0]
    assert np.array_equal(t, to.to_tensor_normalize(ramp))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["eval", "eval_random_views", "train_aug", "flip", "train_occlusion"])
def test_pipeline_matches_oracle(case, tmp_path):
    """shard on disk -> MultiviewWebDataset -> collation, per-frame launches and the single-launch batch form, against the
    oracle's process_data_item on the same records (pixels bit-exact, labels exact)."""
    import poem_v2_amd as pk
    ds, frames, rnd, vr, aug, flip, dt, seed = CASES[case]
    path = str(tmp_path / f"{ds}_mv_test-000000.tar")
    pk.wds.write_shard(path, list(_frames(case)))
    tf = dict(is_train=bool(aug), aug=_oracle_aug(aug))
    random.seed(seed)
    np.random.seed(seed)
    want = [to.process_data_item(pk.wds.decode_record(r), inv_extr=ds in pk.wds.INV_EXTR_DATASETS, random_n_views=rnd,
                                 view_range=vr, **tf) for r in pk.wds.tar_records(path)]
    for defer in (False, True):
        dset = _dataset(case, defer)
        dset.shards = [path]
        dset.is_train = False                                    # keep file order (the augmentation stays on)
        random.seed(seed)
        np.random.seed(seed)
        got = list(dset)
        assert len(got) == len(want)
        batch = pk.collation_random_n_views(got, transform=dset.transform)
        img = batch["image"].cpu().numpy()
        assert img.shape == (sum(len(w["image"]) for w in want), 3, 256, 256)
        assert np.array_equal(img, np.concatenate([w["image"] for w in want]))
        for k in ("target_cam_intr", "target_cam_extr", "target_joints_3d", "master_joints_3d" if False else "affine"):
            assert np.array_equal(batch[k].numpy(), np.concatenate([w[k] for w in want]).astype(np.float32)), k
        assert list(batch["cam_view_num"]) == [len(w["image"]) for w in want]


@pytest.mark.gpu
def test_full_batch_identity_crop_property():
    """BASELINE configs[1] scale (32 frames x 8 views, 640x480 sources): a pure integer translation is an exact crop,
    whatever the batch it runs in (size-independent property; the oracle is not needed)."""
    import poem_v2_amd as pk
    g = np.random.default_rng(9)
    base = [g.integers(0, 256, size=(480, 640, 3), dtype=np.uint8) for _ in range(8)]
    imgs = [base[i % 8] for i in range(256)]
    offs = [(int(g.integers(0, 384)), int(g.integers(0, 224))) for _ in range(256)]
    Ms = [np.array([[1, 0, -ox], [0, 1, -oy]], np.float32) for ox, oy in offs]
    out = pk.transform.warp_views(imgs, Ms, (256, 256), device=DEV, out="u8").cpu().numpy()
    for i in (0, 1, 17, 100, 255):
        ox, oy = offs[i]
        assert np.array_equal(out[i], imgs[i][oy:oy + 256, ox:ox + 256])
    alone = pk.transform.warp_views(imgs[100:101], Ms[100:101], (256, 256), device=DEV, out="u8").cpu().numpy()
    assert np.array_equal(alone[0], out[100])


@pytest.mark.gpu
def test_errors_are_loud():
    import poem_v2_amd as pk
    im = np.zeros((8, 8, 3), np.uint8)
    with pytest.raises(RuntimeError):
        pk.transform.warp_views([im], [np.eye(3)], (8, 8), device="cpu")
    with pytest.raises(ValueError):
        pk.transform.warp_views([im.astype(np.float32)], [np.eye(3)], (8, 8), device=DEV)
    L = pk.hip.lib()
    assert L.poem_warp_affine(None, None, None, None, None, None, None, 1, 8, 8, None) < 0
