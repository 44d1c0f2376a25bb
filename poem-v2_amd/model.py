"""``PtEmbedMultiviewStereoV2`` -- the caller of the hot path, inference side only (lib/models/POEM.py:39-332 upstream):
images -> HRNet pyramid -> ``feat_decode`` / ``heatmap_stage`` -> ragged DLT -> ``POEM_Generalized_Head`` -> the
``preds`` dict the reference's ``_forward_impl`` returns (same keys).

Which part runs where: the backbone is plain PyTorch-ROCm (backbone.py; out of the hot path), everything after it is
the HIP path (decode.py, triangulation.py, head.py -> libpoem_hip.so).  Training-mode noise on the reference joints
(POEM.py:272-281), losses and summaries are not built."""
import numpy as np
import torch

from .backbone import HRNet
from .builder import MODEL, CN, build_head
from .decode import FeatureDecoders
from .triangulation import triangulate_reference_joints


@MODEL.register_module()
class PtEmbedMultiviewStereoV2:

    def __init__(self, cfg, device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("PtEmbedMultiviewStereoV2 runs on the MI355X HIP path only (no CPU fallback)")
        self.name = type(self).__name__
        self.cfg = cfg
        self.device = torch.device(cfg.get("DEVICE", device))
        preset = cfg.get("DATA_PRESET", CN({}))
        self.center_idx = int(preset.get("CENTER_IDX", 9))                       # POEM.py:50
        self.num_joints = 21
        self.img_backbone = HRNet(cfg.get("BACKBONE", None), device=self.device)  # POEM.py:57
        self.ptEmb_head = build_head(cfg.HEAD, data_preset=preset)                # POEM.py:114
        self.num_preds = self.ptEmb_head.num_preds
        self.decoders = None

    # -- weights ----------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd):
        """Full-model checkpoint in the reference's key names: ``img_backbone.*``, ``feat_delayer.*`` / ``feat_in.*`` /
        ``uv_delayer.*`` / ``uv_out.*``, ``ptEmb_head.*``.  Returns the keys that were ignored (dead tensors)."""
        ignored = self.img_backbone.load_state_dict(sd, prefix="img_backbone.")
        self.decoders = FeatureDecoders.load_reference_state_dict(sd, self.device)
        head_sd = {k[len("ptEmb_head."):]: v for k, v in sd.items() if k.startswith("ptEmb_head.")}
        ignored += ["ptEmb_head." + k for k in self.ptEmb_head.load_reference_state_dict(head_sd)]
        self.ptEmb_head.to(self.device).eval()
        return ignored

    def load_parts(self, backbone_sd, decoder_sd, head_sd, template=None):
        self.img_backbone.load_state_dict(backbone_sd)
        self.decoders = FeatureDecoders(decoder_sd, self.device)
        self.ptEmb_head.load_state_dict(head_sd, strict=False)
        if template is not None:
            self.ptEmb_head.set_template(template)
        self.ptEmb_head.to(self.device).eval()
        return self

    # -- forward ----------------------------------------------------------------------------------------------------
    def extract_img_feat(self, img):
        return self.img_backbone(img)                                            # POEM.py:246

    @torch.no_grad()
    def _forward_impl(self, batch, **kwargs):
        """batch: ``image`` (BN,3,H,W), ``target_cam_intr`` (BN,3,3), ``target_cam_extr`` (BN,4,4), ``master_id``,
        ``cam_view_num`` (B,), ``master_joints_3d`` (only read when every sample has one view)   [POEM.py:250-332]"""
        if kwargs.get("mode", "test") == "train":
            raise NotImplementedError("training mode is outside the built path")
        img = batch["image"]
        img = img.view(-1, img.shape[-3], img.shape[-2], img.shape[-1]).to(self.device)
        views = np.asarray(batch["cam_view_num"]).astype(np.int64)
        batch_size, BN = len(views), img.shape[0]
        H, W = img.shape[-2:]
        img_feats = self.extract_img_feat(img)
        mlvl_feat = self.decoders.feat_decode(img_feats, self.img_backbone.name)            # :267
        uv_pred = self.decoders.heatmap_stage(img_feats, W, H)                              # :270
        K = batch["target_cam_intr"].reshape(-1, 3, 3).to(self.device)
        T = batch["target_cam_extr"].reshape(-1, 4, 4).to(self.device)
        if BN == batch_size:                                                                # :273,282-283
            ref_joints = batch["master_joints_3d"].reshape(-1, 21, 3).to(self.device)
        else:
            if views.min() < 2:
                raise ValueError("a batch mixing single-view and multi-view samples has no DLT solution for the former "
                                 "(upstream's SVD returns the null vector of a rank-2 system there)")
            ref_joints = triangulate_reference_joints(uv_pred, K, T, views)                 # :284-299
        img_metas = {"inp_img_shape": (H, W), "cam_intr": K, "cam_extr": T, "master_id": batch["master_id"],
                     "cam_view_num": views}
        preds = self.ptEmb_head(mlvl_feat=mlvl_feat, img_metas=img_metas, reference_joints=ref_joints)
        j = preds["all_coords_preds"][-1, :, :self.num_joints, :]
        v = preds["all_coords_preds"][-1, :, self.num_joints:, :]
        centre = j[:, self.center_idx, :].unsqueeze(1)
        preds.update(pred_joints_3d=j, pred_verts_3d=v, pred_joints_3d_rel=j - centre, pred_verts_3d_rel=v - centre,
                     pred_joints_uv=uv_pred, pred_ref_joints_3d=ref_joints)                 # :321-331
        return preds

    def testing_step(self, batch, step_idx=0, **kwargs):
        return self._forward_impl(batch, mode="test", **kwargs)

    def inference_step(self, batch, step_idx=0, **kwargs):
        return self._forward_impl(batch, mode="inference", **kwargs)

    def forward(self, inputs, step_idx=0, mode="test", **kwargs):                           # POEM.py:486-496
        if mode in ("val", "test"):
            return self.testing_step(inputs, step_idx, **kwargs)
        if mode == "inference":
            return self.inference_step(inputs, step_idx, **kwargs)
        raise ValueError(f"mode {mode} is not built (inference side only)")

    __call__ = forward
