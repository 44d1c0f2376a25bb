"""End-to-end scope plumbing (SURVEY 8d "E2E"): the PyTorch HRNet-W40 backbone pinned to the reference's own HRNet
(CPU), and the model-level caller ``PtEmbedMultiviewStereoV2`` from images to the preds dict (GPU)."""
import json
import os

import numpy as np
import pytest
import torch

import poem_oracle as po
import poem_v2_amd as pk
from poem_v2_amd import backbone as bb
from util import GOLDEN


def _golden():
    z = np.load(os.path.join(GOLDEN, "backbone.npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def test_hrnet_matches_reference_backbone_outputs():
    z, meta = _golden()
    sd = bb.seeded_hrnet_state_dict(meta["seed"])
    assert len(sd) == meta["live_keys"]
    net = bb.HRNet(state_dict=sd)
    img = 0.3 * torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(meta["seed"]))
    ys = net(img)
    assert [tuple(y.shape) for y in ys] == [(2, 40, 16, 16), (2, 80, 8, 8), (2, 160, 4, 4), (2, 320, 2, 2)]
    for i, y in enumerate(ys):
        ref = torch.from_numpy(z[f"level{i}"])
        # BatchNorm folded into the conv weights (fp64 fold): agrees with conv -> BN to fp32 round-off over ~100 layers
        assert float((y - ref).abs().max()) < 2e-5 * float(ref.abs().max()), i


def test_hrnet_key_table_and_prefix_loading():
    shapes = bb.hrnet_param_shapes()
    assert shapes["conv1.weight"] == (64, 3, 3, 3) and shapes["layer1.0.downsample.0.weight"] == (256, 64, 1, 1)
    assert shapes["transition1.1.0.0.weight"] == (80, 256, 3, 3) and shapes["transition3.3.0.0.weight"] == (320, 160, 3, 3)
    assert shapes["stage4.2.fuse_layers.3.0.2.0.weight"] == (320, 40, 3, 3)
    assert shapes["stage3.1.fuse_layers.0.2.0.weight"] == (40, 160, 1, 1)
    sd = {"img_backbone." + k: v for k, v in bb.seeded_hrnet_state_dict(1).items()}
    sd["img_backbone.classifier.weight"] = torch.zeros(1000, 2048)          # dead head: ignored, reported
    net = bb.HRNet()
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 64, 64))
    ignored = net.load_state_dict(sd, prefix="img_backbone.")
    assert ignored == ["img_backbone.classifier.weight"]
    with pytest.raises(KeyError):
        bb.HRNet().load_state_dict({"conv1.weight": torch.zeros(64, 3, 3, 3)})
    assert pk.builder.BACKBONE.get("HRNet") is bb.HRNet and pk.builder.MODEL.get("PtEmbedMultiviewStereoV2") is not None


@pytest.mark.gpu
def test_model_images_to_preds_matches_staged_oracles():
    """images -> HRNet (PyTorch-ROCm) -> feat_decode / heatmap_stage / DLT / head (HIP) against the same chain built from
    the CPU backbone + the decode / DLT / path oracles."""
    import decode_oracle as do
    import dlt_oracle
    from util import oracle_consts, run_oracle
    views = [3, 2]
    b = pk.inputs.synthetic_batch(views, seed=4)
    img = pk.inputs.synthetic_images(sum(views), seed=4)
    cfg = pk.CN({"HEAD": pk.configs.head_cfg(128), "DATA_PRESET": {"CENTER_IDX": 9}})
    model = pk.build_model(pk.CN({"TYPE": "PtEmbedMultiviewStereoV2", **cfg}))
    bsd, dsd = bb.seeded_hrnet_state_dict(0), pk.weights.seeded_decoder_state_dict(0)
    hsd = pk.weights.seeded_state_dict(128, seed=0)
    model.load_parts(bsd, dsd, hsd, template=po.synthetic_template(1234))
    batch = {"image": img, "target_cam_intr": b["img_metas"]["cam_intr"], "target_cam_extr": b["img_metas"]["cam_extr"],
             "master_id": [0] * len(views), "cam_view_num": np.asarray(views)}
    seen, backbone = {}, model.extract_img_feat
    model.extract_img_feat = lambda x: seen.setdefault("pyr", backbone(x))     # the pyramid THIS forward used (MIOpen may
    preds = model(batch, 0, mode="test")                                       # pick other solvers on a later call)
    del model.extract_img_feat
    for k in ("all_coords_preds", "pred_joints_3d", "pred_verts_3d", "pred_joints_3d_rel", "pred_verts_3d_rel",
              "pred_joints_uv", "pred_ref_joints_3d"):
        assert k in preds
    assert tuple(preds["pred_verts_3d"].shape) == (2, 778, 3) and tuple(preds["pred_joints_uv"].shape) == (5, 21, 2)
    # stage by stage, each stage's oracle fed with the device's own input to that stage
    pyr_d = seen["pyr"]
    pyr_c = bb.HRNet(state_dict=bsd)(img)
    for yd, yc in zip(pyr_d, pyr_c):                                   # MIOpen vs CPU convolutions, ~100 layers deep
        assert float((yd.cpu() - yc).abs().max()) < 1e-3 * float(yc.abs().max())
    pyr = [y.cpu() for y in pyr_d]
    dsd_o = {k: v.clone() for k, v in dsd.items()}
    mlvl_d = model.decoders.feat_decode(pyr_d)
    mlvl = do.feat_decode(pyr, dsd_o)
    assert float((mlvl_d.cpu() - mlvl).abs().max()) < 2e-5 * float(mlvl.abs().max())
    hm_d, hm = model.decoders.uv_decode(pyr_d).cpu(), do.uv_decode(pyr, dsd_o)
    e_hm = float((hm_d - hm).abs().max())
    # tests/test_decode.py's bar (2e-5 on logits of order 1) relative to this pyramid's logit scale: the HRNet outputs are
    # not unit-scale, and the rounding of the chained K <= 4320 contractions in front of the sigmoid scales with them
    logit_scale = max(1.0, float(torch.logit(hm.double().clamp(1e-12, 1 - 1e-12)).abs().max()))
    assert e_hm < 2e-5 * logit_scale, (e_hm, logit_scale)
    # the read-out alone (device heat maps through the oracle's expectation): summation order only
    uv_same_hm = do.heatmap_to_uv(hm_d, 256, 256)
    assert float((preds["pred_joints_uv"].cpu() - uv_same_hm).abs().max()) < 5e-4      # pixels
    # whole stage: u = sum(x h) / sum(h), so |du| <= 2 e_hm sum|x - u| / sum(h) <= 2 e_hm * 256 * HW / sum(h); these
    # seeded heat maps are nearly flat (sigmoid of small logits), which is the worst case for that bound
    uv = do.heatmap_stage(pyr, dsd_o, 256, 256)
    bound = 2.0 * e_hm * 256.0 * float((hm.shape[-1] * hm.shape[-2] / hm.sum(dim=(-1, -2))).max()) + 5e-4
    assert float((preds["pred_joints_uv"].cpu() - uv).abs().max()) < min(bound, 2e-2)
    rj = dlt_oracle.triangulate_reference_joints(preds["pred_joints_uv"].cpu(), b["img_metas"]["cam_intr"],
                                                 b["img_metas"]["cam_extr"], views)
    assert float((preds["pred_ref_joints_3d"].cpu() - rj).abs().max()) < 5e-6          # metres
    spec_cfg = po.PathConfig(embed=128, nsample=4096, parametric=False)
    ob = dict(b)
    ob["mlvl_feat"] = mlvl_d.cpu()
    ob["reference_joints"] = preds["pred_ref_joints_3d"].cpu()
    ref = run_oracle(spec_cfg, hsd, oracle_consts(4096), ob)["all_coords_preds"]
    got = preds["all_coords_preds"].cpu()
    mpvpe_mm = float((got[-1, :, 21:] - ref[-1, :, 21:]).norm(dim=-1).mean()) * 1e3
    assert mpvpe_mm < 1e-3, mpvpe_mm
    j = preds["pred_joints_3d"]
    assert torch.equal(preds["pred_verts_3d_rel"], preds["pred_verts_3d"] - j[:, 9:10])
