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


def case_setup(spec):
    """spec: dict(embed, nsample, views, seed, parametric) -> (cfg, weights, consts, batch)."""
    cfg = po.PathConfig(embed=spec["embed"], nsample=spec["nsample"], parametric=spec["parametric"])
    w = pk.weights.seeded_state_dict(spec["embed"], seed=spec["seed"], parametric=spec["parametric"])
    consts = oracle_consts(spec["nsample"])
    batch = synthetic_batch(spec["views"], seed=spec["seed"])
    return cfg, w, consts, batch


def run_oracle(cfg, w, consts, batch, taps=None, hoist=False, anchor_tables=False):
    m = batch["img_metas"]
    mano_fn = po.toy_mano(consts["template"], cfg.center_idx) if cfg.parametric else None
    with torch.no_grad():
        return po.head_forward(w, cfg, consts, batch["mlvl_feat"], m["cam_intr"], m["cam_extr"], m["cam_view_num"],
                               batch["reference_joints"], inp_img_shape=m["inp_img_shape"], taps=taps, hoist=hoist,
                               mano_fn=mano_fn, anchor_tables=anchor_tables)


from poem_v2_amd.configs import head_cfg  # noqa: E402,F401


def build_hip_head(spec, device="cuda:0"):
    """HIP head filled with the same seeded weights / template the oracle and the golden vectors use."""
    head = pk.build_head(head_cfg(spec["embed"], spec["nsample"], spec["parametric"]), data_preset=pk.CN({}))
    sd = pk.weights.seeded_state_dict(spec["embed"], seed=spec["seed"], parametric=spec["parametric"])
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
