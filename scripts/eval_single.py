#!/usr/bin/env python
"""Single-setting evaluation of the hot path -- the drop-in counterpart of the reference's ``scripts/eval_single.py``.

Same command line (``--cfg --dataset --view_min --view_max --model -g --reload -p --draw``) and the same YAML edits
(scripts/eval_single.py:63-86 upstream: dataset URL / epoch size / view range, the four size fields derived from the
model category, PARAMETRIC_OUTPUT for medium_MANO), written back to ``--cfg`` exactly as upstream does.  What differs:

* the reference shells out to ``./ddp_python scripts/eval.py`` (full model: HRNet + heat-map stage + DLT + head).  This
  build owns the head / decoder path only, so the evaluation loop here feeds the head the tensors
  the reference's ``_forward_impl`` would hand it (lib/models/POEM.py:306-332): backbone features, cameras and
  triangulated reference joints;
* the dataset tars are not available offline: the source is the seeded synthetic generator of ``poem_v2_amd.inputs``
  with views per sample drawn from ``[view_min, view_max]`` the way ``collation_random_n_views`` does
  (lib/utils/collation.py:7-25 upstream).  ``--epoch_size`` bounds the run (default: 64 samples, not the dataset's).

One process per GPU: run directly (``-g 0``) or under ``python -m torch.distributed.run --nproc-per-node N``; ranks take
contiguous sample shards and the metric sums meet in one all-reduce (RCCL)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import poem_v2_amd as pk  # noqa: E402
from poem_v2_amd import dist as pdist  # noqa: E402
from poem_v2_amd.metrics import Joint3DPCK, MeanEPE, PAEval, Vert3DPCK  # noqa: E402
from poem_v2_amd.triangulation import triangulate_reference_joints  # noqa: E402

# scripts/eval_single.py:5-36 upstream (urls kept for the record; the tars are not shipped)
DATASET_META = {
    "HO3D": {"url": "data/dataset_tars/HO3D_mv_test/HO3D_mv_test-{000000..000002}.tar", "max_view": 5, "epoch_size": 2706},
    "DexYCB": {"url": "data/dataset_tars/DexYCB_mv/DexYCB_mv_test-{000000..000003}.tar", "max_view": 8, "epoch_size": 4950},
    "Arctic": {"url": "data/dataset_tars/Arctic_mv/Arctic_mv_val_p1-{000000..000045}.tar", "max_view": 8, "epoch_size": 17392},
    "Interhand": {"url": "data/dataset_tars/Interhand_mv/Interhand_mv_val-{000000..000022}.tar", "max_view": 8, "epoch_size": 85255},
    "Oakink": {"url": "data/dataset_tars/Oakink_mv/Oakink_mv_test-{000000..000045}.tar", "max_view": 4, "epoch_size": 21351},
    "Freihand": {"url": "data/dataset_tars/Freihand_mv/Freihand_mv_test-{000000..000000}.tar", "max_view": 1, "epoch_size": 3960},
}
MODEL_CATEGORY = ["small", "medium", "large", "huge", "medium_MANO"]
EMBED_SIZE = [128, 256, 512, 1024, 256]


def edit_cfg(cfg, dataset, model_type, view_range):
    """The in-place YAML edits of scripts/eval_single.py:63-86 upstream.  Returns the (possibly adjusted) view range."""
    if dataset not in DATASET_META:
        raise AssertionError(f"Dataset {dataset} not found in dataset_info.")
    if model_type not in MODEL_CATEGORY:
        raise AssertionError(f"Model category {model_type} not found in model_category.")
    meta = DATASET_META[dataset]
    test = cfg.setdefault("DATASET", {}).setdefault("TEST", {})
    tgt = test.setdefault("TARGET", {})
    tgt["URLS"] = meta["url"]
    test["EPOCH_SIZE"] = meta["epoch_size"]
    tgt["EPOCH_SIZE"] = meta["epoch_size"]
    if dataset == "Freihand":
        view_range = [1, 1]
        print("Setting view range to 1 for Freihand dataset.")
    tgt["VIEW_RANGE"] = list(view_range)
    embed = EMBED_SIZE[MODEL_CATEGORY.index(model_type)]
    head = cfg.setdefault("MODEL", {}).setdefault("HEAD", {})
    head.setdefault("POSITIONAL_ENCODING", {})["NUM_FEATS"] = embed // 2
    head.setdefault("TRANSFORMER", {})["INPUT_FEAT_DIM"] = embed
    head["POINTS_FEAT_DIM"] = embed
    head["EMBED_DIMS"] = embed
    head["TRANSFORMER"]["PARAMETRIC_OUTPUT"] = model_type == "medium_MANO"
    return view_range


def default_cfg():
    """A config tree with the release MODEL.HEAD subtree (config/release/train_medium.yaml:185-225 upstream)."""
    def plain(node):
        return {k: plain(v) if isinstance(v, dict) else v for k, v in dict(node).items()}
    head = plain(pk.configs.head_cfg(256))
    head.pop("MAX_VIEWS", None)
    return {"DATASET": {"TEST": {"TARGET": {}}}, "MODEL": {"TYPE": "PtEmbedMultiviewStereoV2", "HEAD": head}}


def random_views(n_samples, view_range, seed):
    rng = np.random.RandomState(seed)
    lo, hi = view_range
    return rng.randint(lo, hi + 1, size=n_samples).tolist()


def load_template(path):
    """(799,3) zero-pose hand template in metres (rows 0..20 joints, 21..798 vertices, centred at joint 9) from a ``.npy`` /
    ``.pt`` file: on a licensed machine, ManoLayer's zero-pose output (lib/models/heads/ptEmb_head.py:886-892 upstream)."""
    t = np.load(path) if path.endswith(".npy") else torch.load(path, map_location="cpu")
    return torch.as_tensor(np.asarray(t), dtype=torch.float32).reshape(799, 3)


def install_template(head, reload, template):
    """Which hand template the head runs with, and a word for the result record.  A seeded synthetic template is only
    right for seeded synthetic weights: with ``--reload`` the template must come from ``--template``; otherwise the head's
    own "synthetic template" warning is left armed and the record says so."""
    if template:
        head.set_template(load_template(template))
        return f"file:{template}"
    if reload:
        return "synthetic (NO --template given with --reload: metrics are not meaningful for a real checkpoint)"
    head.set_template(pk.inputs.synthetic_template(1234))
    return "synthetic(seed=1234)"


def evaluate(cfg, view_range, model_type, device, reload=None, epoch_size=64, batch_size=2, seed=0, verbose=True,
             pyramid=False, template=None):
    rank, _, world = pdist.env_world()
    head_node = pk.CN(cfg["MODEL"]["HEAD"])
    head_node["MAX_VIEWS"] = max(10, int(view_range[1]))
    head = pk.build_head(head_node, data_preset=pk.CN({}))
    embed = head.embed_dims
    if reload:
        sd = torch.load(reload, map_location="cpu")
        sd = sd.get("state_dict", sd.get("model", sd)) if isinstance(sd, dict) else sd
        ignored = head.load_reference_state_dict(sd)
        if verbose and rank == 0:
            print(f"reloaded {reload}: {len(ignored)} dead tensors ignored")
    else:
        head.load_state_dict(pk.weights.seeded_state_dict(embed, seed=0, parametric=head.parametric_output), strict=False)
    template_source = install_template(head, reload, template)
    if head.parametric_output:
        raise SystemExit("medium_MANO needs a MANO layer (licence-gated assets): call head.set_mano_layer(fn) from "
                         "Python; this script evaluates the non-parametric categories")
    head = head.to(device).eval()
    # --pyramid: start one stage earlier, at the backbone's multi-level features (lib/models/POEM.py:264-270 upstream):
    # feat_decode gives the head's mlvl_feat, heatmap_stage the per-view 2-D joints that are triangulated
    decoders = None
    if pyramid:
        if reload:
            decoders = pk.decode.FeatureDecoders.load_reference_state_dict(sd, device)
        else:
            decoders = pk.decode.FeatureDecoders(pk.weights.seeded_decoder_state_dict(0), device)
    views_all = random_views(epoch_size, view_range, seed)
    lo, hi = pdist.shard_by_views(views_all, rank, world)
    mpvpe, mpjpe = MeanEPE("verts", device=device), MeanEPE("joints", device=device)
    pa = PAEval(None, mesh_score=True, device=device)                       # lib/models/POEM.py:147 upstream
    pck = dict(VAL_MIN=0.0, VAL_MAX=0.02, STEPS=20)                         # AUCCallback defaults, lib/utils/testing.py:31
    pck_j, pck_v = Joint3DPCK(device=device, EVAL_TYPE="joints_3d", **pck), Vert3DPCK(device=device, EVAL_TYPE="verts_3d", **pck)
    n_done, t0 = 0, None
    with torch.no_grad():
        for it, s in enumerate(range(lo, hi, batch_size)):
            views = views_all[s:min(s + batch_size, hi)]
            b = pk.inputs.synthetic_batch(views, seed=seed * 100003 + s)
            metas = dict(b["img_metas"])
            metas["cam_intr"], metas["cam_extr"] = metas["cam_intr"].to(device), metas["cam_extr"].to(device)
            # reference joints the way the full model gets them (lib/models/POEM.py:284-299): DLT over each sample's views
            # of the per-view 2-D joints (here: the synthetic joints projected into every view + 1 px noise)
            vs = torch.repeat_interleave(torch.arange(len(views)), torch.tensor(views))
            T = torch.linalg.inv(b["img_metas"]["cam_extr"])
            X = b["reference_joints"][vs]
            pc = (T[:, None, :3, :3] @ X[..., None]).squeeze(-1) + T[:, None, :3, 3]
            q = (b["img_metas"]["cam_intr"][:, None] @ pc[..., None]).squeeze(-1)
            gn = torch.Generator().manual_seed(seed * 31 + s)
            uv = (q[..., :2] / q[..., 2:] + torch.randn(q[..., :2].shape, generator=gn)).to(device)
            mlvl_feat = b["mlvl_feat"].to(device)
            if decoders is not None:
                pyr = [f.to(device) for f in pk.inputs.synthetic_pyramid(sum(views), seed=seed * 100003 + s)]
                mlvl_feat = decoders.feat_decode(pyr)                                   # POEM.py:267
                uv_pred = decoders.heatmap_stage(pyr, 256, 256)                         # POEM.py:270
                # random features carry no hand: keep the pipeline honest (the kernels run, their output is consumed)
                # but triangulate a blend that stays near the synthetic joints so the head sees sane geometry
                uv = uv + 1e-3 * (uv_pred - uv_pred.mean(dim=1, keepdim=True))
            if min(views) >= 2:
                rj = triangulate_reference_joints(uv, metas["cam_intr"], metas["cam_extr"], views)
            else:                                        # single-view samples take the given joints (POEM.py:282-283)
                rj = b["reference_joints"].to(device)
            preds = head(mlvl_feat, metas, rj)["all_coords_preds"]
            g = torch.Generator().manual_seed(seed * 7919 + s)
            gt = (b["reference_joints"][:, 9:10] + 0.05 * torch.randn(len(views), 799, 3, generator=g)).to(device)
            mpjpe.feed(preds[-1, :, :21], gt[:, :21])      # lib/models/POEM.py:443-444 upstream: joints then verts
            mpvpe.feed(preds[-1, :, 21:], gt[:, 21:])
            pa.feed(preds[-1, :, :21], gt[:, :21], preds[-1, :, 21:], gt[:, 21:])
            pck_j.feed({"pred_joints_3d": preds[-1, :, :21]}, {"master_joints_3d": gt[:, :21]})
            pck_v.feed({"pred_verts_3d": preds[-1, :, 21:]}, {"master_verts_3d": gt[:, 21:]})
            if it == 0:                                    # first batch builds the engine; time from the second on
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
            else:
                n_done += len(views)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0 if t0 else 0.0
    mpvpe.reduce(), mpjpe.reduce(), pa.reduce(), pck_j.reduce(), pck_v.reduce()
    pam = pa.get_measures()
    res = {"dataset_source": "synthetic", "scope": "pyramid->verts" if pyramid else "mlvl_feat->verts", "model": model_type, "embed": embed, "view_range": list(view_range),
           "samples": int(mpvpe.acc[1].item()), "MPVPE_mm_vs_synthetic_gt": mpvpe.result() * 1e3,
           "MPJPE_mm_vs_synthetic_gt": mpjpe.result() * 1e3, "PA_MPJPE_mm": pam["pa_mpjpe"] * 1e3,
           "PA_MPVPE_mm": pam["pa_mpvpe"] * 1e3, "auc_j": pck_j.get_measures()["auc_all"], "auc_v": pck_v.get_measures()["auc_all"],
           "samples_per_s_rank0": (n_done / dt) if dt > 0 and n_done else None, "world_size": world,
           "template_source": template_source}
    return res


def evaluate_shards(cfg, view_range, model_type, device, shard_dir, dataset, reload=None, epoch_size=16, batch_size=2,
                    n_cams=8, raw_size=(640, 480), template=None):
    """Images -> metrics from record shards (SURVEY 8f N4 in front of the model): ``MultiviewWebDataset`` over the URLS of the
    edited config (tar records: ``image_<i>.png|jpg`` + ``label.pyd``), the per-view crop / warp / normalise on the
    device (one launch per batch), ``collation_random_n_views``, then the model-level caller
    (``PtEmbedMultiviewStereoV2``: HRNet on PyTorch-ROCm -> decode / heat maps / DLT / head on HIP) and the device metrics
    against the records' ``master_joints_3d`` / ``master_verts_3d`` (lib/models/POEM.py:596-610 upstream).
    The dataset tars are not available offline: when ``shard_dir`` holds no shard of the dataset's name, seeded synthetic
    shards of the same record layout are written there first."""
    rank, _, world = pdist.env_world()
    ds_node = cfg["DATASET"]["TEST"]["TARGET"]
    pattern = os.path.join(shard_dir, os.path.basename(ds_node["URLS"]))
    urls = pk.wds.expand_urls(pattern)
    if rank == 0 and not all(os.path.exists(u) for u in urls):
        synthetic_frame = pk.inputs.synthetic_frame
        os.makedirs(shard_dir, exist_ok=True)
        per = -(-epoch_size // len(urls))
        for si, u in enumerate(urls):
            pk.wds.write_shard(u, [synthetic_frame(si * per + i, n_cams=n_cams, raw=raw_size) for i in range(per)])
    pdist.barrier()
    node = pk.wds.dataset_cfg(pattern, view_range=view_range, device=str(device))
    dset = pk.MultiviewWebDataset(node, data_preset=node.DATA_PRESET, is_train=False, defer_images=True, rank=rank, world=world)
    head_node = dict(cfg["MODEL"]["HEAD"])
    head_node["MAX_VIEWS"] = max(10, int(view_range[1]))
    model = pk.build_model(pk.CN({"TYPE": "PtEmbedMultiviewStereoV2", "HEAD": head_node, "DATA_PRESET": {"CENTER_IDX": 9},
                                  "DEVICE": str(device)}))
    if reload:
        sd = torch.load(reload, map_location="cpu")
        model.load_state_dict(sd.get("state_dict", sd.get("model", sd)) if isinstance(sd, dict) else sd)
        template_source = install_template(model.ptEmb_head, reload, template)
    else:
        template_source = f"file:{template}" if template else "synthetic(seed=1234)"
        from poem_v2_amd.backbone import seeded_hrnet_state_dict
        model.load_parts(seeded_hrnet_state_dict(0), pk.weights.seeded_decoder_state_dict(0),
                         pk.weights.seeded_state_dict(model.ptEmb_head.embed_dims, seed=0),
                         template=load_template(template) if template else pk.inputs.synthetic_template(1234))
    mpvpe, mpjpe = MeanEPE("verts", device=device), MeanEPE("joints", device=device)
    n, t0, frames = 0, None, []

    def run(frames):
        batch = pk.collation_random_n_views(frames, transform=dset.transform)
        if min(batch["cam_view_num"]) < 2 <= max(batch["cam_view_num"]):
            return 0                                         # upstream has no DLT answer for such a mix either
        preds = model(batch, 0, mode="test")
        gj = batch["master_joints_3d"].reshape(-1, 21, 3).to(device)     # the collation concatenates per-frame (21,3) arrays
        gv = batch["master_verts_3d"].reshape(-1, 778, 3).to(device)
        mpjpe.feed(preds["pred_joints_3d"], gj)
        mpvpe.feed(preds["pred_verts_3d"], gv)
        return len(frames)

    for f in dset:
        frames.append(f)
        if len(frames) == batch_size:
            done = run(frames)
            frames = []
            if t0 is None:
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
            else:
                n += done
    if frames:
        n += run(frames)
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0 if t0 else 0.0
    mpvpe.reduce(), mpjpe.reduce()
    return {"dataset_source": f"record shards {pattern} (synthetic records)", "scope": "shards->images->verts",
            "model": model_type, "view_range": list(view_range), "samples": int(mpvpe.acc[1].item()),
            "MPVPE_mm_vs_record_gt": mpvpe.result() * 1e3, "MPJPE_mm_vs_record_gt": mpjpe.result() * 1e3,
            "samples_per_s_rank0": (n / dt) if dt > 0 and n else None, "world_size": world,
            "template_source": template_source}


def main(args):
    view_range = [args.view_min, args.view_max]
    if args.cfg and os.path.exists(args.cfg):
        with open(args.cfg, "r") as f:
            cfg = yaml.load(f, Loader=yaml.FullLoader)
    else:
        cfg = default_cfg()
    view_range = edit_cfg(cfg, args.dataset, args.model, view_range)
    # upstream dumps the edited tree back to the same path from its single launcher process; under torch.distributed.run
    # every rank executes this script, so only rank 0 writes (via a rename: a concurrent reader sees the old or the new
    # file, never a truncated one) and the others keep their in-memory edit
    if args.cfg and int(os.environ.get("RANK", "0")) == 0:
        tmp = f"{args.cfg}.tmp{os.getpid()}"
        with open(tmp, "w") as f:
            yaml.dump(cfg, f)
        os.replace(tmp, args.cfg)
    rank, local_rank, world = pdist.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("eval_single.py needs a GPU: the head runs on the MI355X HIP path only (no CPU fallback)")
    device = torch.device("cuda", local_rank if world > 1 else args.gpu_id)
    torch.cuda.set_device(device)
    if args.draw and rank == 0:
        print("--draw: rendering is outside the hot path and not built (DESIGN.md section 0); metrics only")
    if args.shards:
        res = evaluate_shards(cfg, view_range, args.model, device, args.shards, args.dataset, reload=args.reload,
                              epoch_size=args.epoch_size, batch_size=args.batch_size, template=args.template)
    else:
        res = evaluate(cfg, view_range, args.model, device, reload=args.reload, epoch_size=args.epoch_size,
                       batch_size=args.batch_size, pyramid=args.pyramid, template=args.template)
    if rank == 0:
        exp_id = f"{args.dataset}_view_{view_range[0]}_{view_range[1]}_{args.model}"
        if args.model == "huge":
            # C = 1024 (dh = 256) is outside the widths the fused kernels are instantiated for (merge.hip / chain.hip: 128 / 256 /
            # 512): the head runs the round-1 operator sequence -- same results, ~100 samples/s at batch 8 instead of the
            # medium model's 1400 (not a BASELINE config; upstream's README lists only the other four models)
            res["note"] = "POEM-huge runs the operator sequence (no fused front end / chain kernels at embed 1024)"
        print(json.dumps({"exp_id": exp_id, **res}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    parser = argparse.ArgumentParser(description="Eval Single Setting")
    parser.add_argument("--cfg", type=str, required=True, help="Path to the configuration file.")
    parser.add_argument("--dataset", type=str, required=True, help="Dataset name.")
    parser.add_argument("--view_min", type=int, required=True, help="Minimum view range.")
    parser.add_argument("--view_max", type=int, required=True, help="Maximum view range.")
    parser.add_argument("--model", type=str, required=True, help="Model category.")
    parser.add_argument("--gpu_id", "-g", type=int, default=0, required=True, help="GPU ID to run the evaluation.")
    parser.add_argument("--reload", type=str, default=None, help="Path to the checkpoint to reload.")
    parser.add_argument("--port", "-p", type=int, default=60000, help="Port to run the evaluation.")
    parser.add_argument("--draw", "-d", action="store_true", help="Visualize the results.")
    parser.add_argument("--epoch_size", type=int, default=64, help="Synthetic samples to evaluate (this build).")
    parser.add_argument("--pyramid", action="store_true",
                        help="start at the backbone's multi-level features: feat_decode + heatmap_stage on HIP (this build).")
    parser.add_argument("--shards", type=str, default=None, metavar="DIR",
                        help="evaluate the full model from record shards under DIR (the dataset's URLS pattern; synthetic "
                             "shards are written there when absent): tar records -> device transform -> model -> metrics")
    parser.add_argument("--template", type=str, default=None, metavar="FILE",
                        help="(799,3) zero-pose hand template (.npy / .pt; ManoLayer's zero-pose joints + vertices, centred at "
                             "joint 9).  Required for meaningful metrics with --reload; synthetic otherwise (this build).")
    parser.add_argument("--batch_size", type=int, default=2, help="--val_batch_size of the reference (lib/opt.py:27-30).")
    main(parser.parse_args())
