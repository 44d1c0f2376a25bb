"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os

import numpy as np
import torch

import poem_oracle as po
import poem_v2_amd as pk
from poem_v2_amd.inputs import synthetic_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
ASSETS = os.path.join(ROOT, "poem-v2_amd", "assets")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    return z, meta


def oracle_consts(nsample=4096):
    bps = torch.from_numpy(np.load(os.path.join(ASSETS, "bps.npy")))[0, :nsample].contiguous()
    anchor = torch.from_numpy(np.load(os.path.join(ASSETS, "anchor.npy")))[0].contiguous()
    anchor_idx = torch.from_numpy(np.load(os.path.join(ASSETS, "anchor_idx.npy")))[0].contiguous()
    return dict(bps=bps, anchor=anchor, anchor_idx=anchor_idx, template=po.synthetic_template(1234))


def seeded_weights(spec):
    extra = dict(petr=True, depth_num=spec.get("depth_num", 32)) if spec.get("petr") else {}
    if "nblocks" in spec:
        extra["nblocks"] = spec["nblocks"]
    return pk.weights.seeded_state_dict(spec["embed"], seed=spec["seed"], parametric=spec["parametric"],
                                        gain=spec.get("gain", 1.0), ln_spread=spec.get("ln_spread", 0.02), **extra)


PE_SPEC_KEYS = ("pe_normalize", "petr", "depth_num", "lid", "depth_start", "depth_end", "position_range",      # round 5
                "knn", "knn_query", "heads", "nblocks")                                                         # round 6


def case_setup(spec):
    """spec: dict(embed, nsample, views, seed, parametric) -> (cfg, weights, consts, batch)."""
    cfg = po.PathConfig(embed=spec["embed"], nsample=spec["nsample"], parametric=spec["parametric"],
                        **{k: (tuple(spec[k]) if k == "position_range" else spec[k]) for k in PE_SPEC_KEYS if k in spec})
    w = seeded_weights(spec)
    consts = oracle_consts(spec["nsample"])
    batch = synthetic_batch(spec["views"], seed=spec["seed"], nan_views=spec.get("nan_views"))
    return cfg, w, consts, batch


# The oracle is a torch-CPU restatement: 3-30 s per release-shape case, and several tests evaluate it on the SAME seeded case
# (final vertices, stage taps, both block-0 forms).  Results are kept per process, keyed by the CONTENT of everything the oracle
# reads (so a test that edits a camera or a feature map gets its own entry): the GPU suite spends its time on the GPU path
# instead of re-running the checker (round 5: 776 s of the driver's 1200 s limit).
_ORACLE_CACHE = {}


def _oracle_key(cfg, w, consts, batch, hoist, anchor_tables):
    import hashlib
    import struct
    h = hashlib.sha1()
    m = batch["img_metas"]
    h.update(repr((cfg, bool(hoist), bool(anchor_tables), [int(v) for v in m["cam_view_num"]], tuple(m["inp_img_shape"]))).encode())
    for t in (batch["mlvl_feat"], m["cam_intr"], m["cam_extr"], batch["reference_joints"], consts["bps"], consts["anchor"],
              consts["anchor_idx"], consts["template"]):
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    for k in sorted(w):
        t = w[k].detach().double()
        h.update(k.encode())
        h.update(struct.pack("dd", float(t.sum()), float(t.abs().sum())))
    return h.hexdigest()


def run_oracle(cfg, w, consts, batch, taps=None, hoist=False, anchor_tables=False, stop_after_sampling=False, cache=True):
    """``stop_after_sampling``: only the sampling stage's taps are wanted (x, uv, g, bps_feat, pt_xyz, query_xyz) -- the decoder
    blocks, 90 % of the oracle's time, are skipped and None is returned."""
    m = batch["img_metas"]
    mano_fn = po.toy_mano(consts["template"], cfg.center_idx) if cfg.parametric else None
    # (cache=False: a test that swaps one of the oracle's own functions for an experiment; the key also carries the identity of
    #  the functions tests are known to replace)
    key = _oracle_key(cfg, w, consts, batch, hoist, anchor_tables) + f"|{id(po.linear)}|{id(po.knn_distances)}"
    hit = _ORACLE_CACHE.get(key) if cache else None
    if hit is not None and (taps is None or hit[1] is not None):
        if taps is not None:
            taps.update(hit[1])
        return None if stop_after_sampling else dict(hit[0])
    if stop_after_sampling:
        assert taps is not None
        taps["__stop_after_sampling__"] = True
    # (taps are recorded whenever they are small enough to keep: the next test of the same case asks for them)
    small = batch["mlvl_feat"].shape[0] * cfg.embed * cfg.nsample * 4 < (256 << 20)
    rec = taps if taps is not None else ({} if small else None)
    with torch.no_grad():
        out = po.head_forward(w, cfg, consts, batch["mlvl_feat"], m["cam_intr"], m["cam_extr"], m["cam_view_num"],
                              batch["reference_joints"], inp_img_shape=m["inp_img_shape"], taps=rec, hoist=hoist,
                              mano_fn=mano_fn, anchor_tables=anchor_tables)
    if out is None:
        rec.pop("__stop_after_sampling__", None)
        return None
    if cache:
        _ORACLE_CACHE[key] = (dict(out), dict(rec) if (rec is not None and small) else None)
    return out


from poem_v2_amd.configs import head_cfg  # noqa: E402,F401


def build_hip_head(spec, device="cuda:0"):
    """HIP head filled with the same seeded weights / template the oracle and the golden vectors use."""
    hc = head_cfg(spec["embed"], spec["nsample"], spec["parametric"])
    hc["POSITIONAL_ENCODING"]["NORMALIZE"] = bool(spec.get("pe_normalize", True))
    hc["PETR_EMBEDDING"] = bool(spec.get("petr", False))
    hc["TRANSFORMER"]["N_BLOCKS"] = spec.get("nblocks", 3)
    hc["TRANSFORMER"]["NUM_ATTENTION_HEADS"] = spec.get("heads", 4)
    hc["TRANSFORMER"]["N_NEIGHBOR"] = spec.get("knn", 32)
    hc["TRANSFORMER"]["N_NEIGHBOR_QUERY"] = spec.get("knn_query") or spec.get("knn", 32)
    for key, name in (("DEPTH_NUM", "depth_num"), ("LID", "lid"), ("DEPTH_START", "depth_start"), ("DEPTH_END", "depth_end"),
                      ("POSITION_RANGE", "position_range")):
        if name in spec:
            hc[key] = spec[name]
    head = pk.build_head(hc, data_preset=pk.CN({}))
    sd = seeded_weights(spec)
    missing, unexpected = head.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    head.set_template(po.synthetic_template(1234))
    head = head.to(device).eval()
    if spec["parametric"]:
        tmpl = head.template
        fn = po.toy_mano(tmpl, 9)
        head.set_mano_layer(fn)
    return head


def batch_to(batch, device):
    m = dict(batch["img_metas"])
    m["cam_intr"] = m["cam_intr"].to(device)
    m["cam_extr"] = m["cam_extr"].to(device)
    return batch["mlvl_feat"].to(device), m, batch["reference_joints"].to(device)


# ---- stage-level comparison against a release-shape fixture (tests + tools/parity_report.py) -----------------------------
TAP_KEYS = ("h_cross", "f_self", "f_cross", "feats")


def reference_neighbours(z, block, which, pt_xyz):
    """Neighbour ids (B,799,32) the reference used in decoder block ``block`` (1 or 2; block 0 takes the fixed anchors): the
    tap the generator records in every release-shape fixture (tests/golden/make_golden.py; round 5: the five benign release
    fixtures were regenerated with it -- nothing is recomputed here any more)."""
    key = f"tap.b{block}.idx_{which}"
    assert key in z.files, f"fixture lacks {key}: regenerate it with tests/golden/make_golden.py"
    return torch.from_numpy(z[key].astype(np.int64))


def neighbour_report(z, block, which, got_idx, pt_xyz):
    """How the neighbour sets of ``got_idx`` (B,799,32) relate to the reference's in one block / attention.
    -> dict(set_equal = fraction of queries with the identical SET, flips = [(sample, query, rel_gap)]) where rel_gap is
    the relative distance gap between the reference's 32nd and 33rd candidate for that query: a flip is *attributed* to a
    near-tie when that gap is at fp32 round-off level."""
    ref = reference_neighbours(z, block, which, pt_xyz)
    got = torch.as_tensor(got_idx).long().reshape(ref.shape)
    same = (torch.sort(got, dim=-1).values == torch.sort(ref, dim=-1).values).all(-1)
    flips = []
    if not bool(same.all()):
        xyz = torch.from_numpy(z[f"tap.b{block - 1}.xyz"])
        src = xyz if which == "self" else pt_xyz
        for b, q in torch.nonzero(~same).tolist():
            d = xyz[b, q][None] - src[b]
            d = d * d
            sd = torch.sort((d[:, 0] + d[:, 1]) + d[:, 2]).values
            flips.append((b, q, float((sd[32] - sd[31]) / sd[31])))
    return {"set_equal": float(same.float().mean()), "flips": flips, "same_mask": same}


def stage_report(z, spec, tap, oracle_taps):
    """``tap(name, shape)`` -> tensor of the path under test.  Per block and stage: max |path - reference| on the fixture's
    strided rows, the oracle's own distance on the same rows and the tensor's scale -- once over all stored rows and once
    over the *clean* rows only (queries whose neighbour sets, in this and every earlier block, are the reference's).
    Per block and attention: the neighbour-set report.  Shared by tests/test_hip_parity.py and tools/parity_report.py."""
    B, C, Q = len(spec["views"]), spec["embed"], 799
    pt_xyz = oracle_taps["pt_xyz"]
    rep = {"stages": {}, "neighbours": {}}
    clean = torch.ones(B, Q, dtype=torch.bool)
    for i in range(3):
        if i > 0:
            for which in ("self", "cross"):
                nb = neighbour_report(z, i, which, tap(f"b{i}.idx_{which}", (B, Q, 32), torch.int32), pt_xyz)
                clean &= nb.pop("same_mask")
                rep["neighbours"][f"b{i}.{which}"] = nb
        for k in TAP_KEYS + ("xyz",):
            ref = torch.from_numpy(z[f"tap.b{i}.{k}"])
            got = tap(f"b{i}.{k}", (B, Q, 3 if k == "xyz" else C)).cpu()
            orc = oracle_taps[f"b{i}.{k}"]
            step = 1 if k == "xyz" else (Q + ref.shape[1] - 1) // ref.shape[1]
            got, orc, rows = got[:, ::step], orc[:, ::step], clean[:, ::step]
            assert got.shape == ref.shape, (k, got.shape, ref.shape)
            dg, do = (got - ref).abs().amax(-1), (orc - ref).abs().amax(-1)          # per stored row
            rep["stages"][f"b{i}.{k}"] = {
                "scale": float(ref.abs().max()), "path_all": float(dg.max()), "oracle_all": float(do.max()),
                "path_clean": float(dg[rows].max()) if bool(rows.any()) else 0.0,
                "oracle_clean": float(do[rows].max()) if bool(rows.any()) else 0.0, "clean_rows": float(rows.float().mean())}
    return rep
