"""``PtEmbedTRv4`` -- the point-embedded transformer decoder, MI355X-native.

Drop-in for the reference class of the same name (lib/models/layers/ptEmb_transformer.py:303-376): same registry
name, constructor config keys, ``forward(query_xyz, query_feat, pt_xyz, pt_feats)`` signature/returns and the same
``state_dict`` key names for every tensor the forward reads (the modules below are parameter containers only -- the
arithmetic runs in libpoem_hip.so).  Dead tensors of the reference (BERT word embeddings, pooler, position
embeddings) are not instantiated."""
import json
import os

import torch
import torch.nn as nn

from .builder import TRANSFORMER


class _Bag(nn.Module):
    """Named container (no forward)."""


def _mlp(i, h, o):
    return nn.Sequential(nn.Linear(i, h), nn.ReLU(), nn.Linear(h, o))


def _bert_attention(C, eps):
    a = _Bag()
    a.self = _Bag()
    a.self.query, a.self.key, a.self.value = nn.Linear(C, C), nn.Linear(C, C), nn.Linear(C, C)
    a.output = _Bag()
    a.output.dense = nn.Linear(C, C)
    a.output.LayerNorm = nn.LayerNorm(C, eps=eps)
    return a


def _vec_attn(C):
    v = _Bag()
    v.fc1, v.fc2 = nn.Linear(C, C), nn.Linear(C, C)
    v.fc_delta = _mlp(3, C, C)
    v.fc_gamma = _mlp(C, C, C)
    v.w_qs, v.w_ks, v.w_vs = nn.Linear(C, C, bias=False), nn.Linear(C, C, bias=False), nn.Linear(C, C, bias=False)
    return v


def _block(C, eps, parametric, nquery):
    b = _Bag()
    b.embedding = nn.Linear(C, C)
    b.encoder = _Bag()
    b.encoder.attn = _bert_attention(C, eps)
    b.encoder.cross_attn = _bert_attention(C, eps)
    b.encoder.vec_attn = _Bag()
    b.encoder.vec_attn.reg_branch = _mlp(C, C, 3)
    b.encoder.vec_attn.query_self_attn = _vec_attn(C)
    b.encoder.vec_attn.query_cross_attn = _vec_attn(C)
    b.encoder.intermediate = _Bag()
    b.encoder.intermediate.dense = nn.Linear(C, 4 * C)
    b.encoder.output = _Bag()
    b.encoder.output.dense = nn.Linear(4 * C, C)
    b.encoder.output.LayerNorm = nn.LayerNorm(C, eps=eps)
    if parametric:
        b.flat_verts = nn.Linear(nquery, 1)
        b.mano_linear = nn.Linear(C, 106)
    return b


@TRANSFORMER.register_module()
class PtEmbedTRv4(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        self.name = type(self).__name__
        self.cfg = cfg
        self.input_feat_dim = cfg.INPUT_FEAT_DIM
        self.dropout = cfg.DROPOUT                     # identity at inference; kept for config parity
        self.num_hidden_layers = cfg.NUM_HIDDEN_LAYERS  # unused by the arithmetic (as upstream)
        self.num_attention_heads = cfg.NUM_ATTENTION_HEADS
        self.bps_feature_dim = cfg.BPS_FEAT_DIM
        self.parametric_output = cfg.get("PARAMETRIC_OUTPUT", False)
        self.mano_center_idx = cfg.get("TRANSFORMER_CENTER_IDX", 9)
        self.nneighbor = cfg.N_NEIGHBOR
        self.nneighbor_query = cfg.N_NEIGHBOR_QUERY
        self.layer_num = cfg.N_BLOCKS
        self.nquery = 799
        if not (1 <= self.nneighbor <= 32 and 1 <= self.nneighbor_query <= 32):
            raise NotImplementedError("the MI355X vector attention holds 32 neighbour columns per query: N_NEIGHBOR and "
                                      "N_NEIGHBOR_QUERY must be in 1..32 (counts below 32 run the masked kernel)")
        # BertConfig defaults the reference reads from config/backbone/bert_cfg.json (ptEmb_transformer.py:334)
        self.initializer_range, self.layer_norm_eps = 0.02, 1e-12
        p = os.path.join("config", "backbone", "bert_cfg.json")
        if os.path.exists(p):
            with open(p) as f:
                j = json.load(f)
            self.initializer_range = j.get("initializer_range", 0.02)
            self.layer_norm_eps = j.get("layer_norm_eps", 1e-12)
        C = self.input_feat_dim
        assert C % self.num_attention_heads == 0
        self.pt_metro_encoder = nn.ModuleList(
            [_block(C, self.layer_norm_eps, self.parametric_output, self.nquery) for _ in range(self.layer_num)])
        self._init_weights()
        self._engine_owner = None   # the head that owns the HIP engine (set by POEM_Generalized_Head)
        self._own_engine = None

    def _init_weights(self):
        # BERT-v4 ``init_weights`` as applied by point_METRO_block to every sub-module (pt_metro_transformer.py:129)
        for m in self.pt_metro_encoder.modules():
            if isinstance(m, nn.Linear):
                m.weight.data.normal_(mean=0.0, std=self.initializer_range)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, nn.LayerNorm):
                m.bias.data.zero_()
                m.weight.data.fill_(1.0)

    def _standalone_engine(self, device):
        """Engine for direct use of the decoder without a head: head-level tensors are zeros."""
        from . import hip
        from .weights import live_key_shapes
        C = self.input_feat_dim
        sig = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._own_engine is None or self._own_engine[0] != sig:
            sd = {"transformer." + k: v for k, v in self.state_dict().items()}
            shapes = live_key_shapes(C, 160, self.nquery, self.layer_num, self.parametric_output)
            w = {k: (sd[k] if k in sd else torch.zeros(s)) for k, s in shapes.items()}
            cfg = hip.make_config(C, heads=self.num_attention_heads, nblocks=self.layer_num,
                                  parametric=self.parametric_output, ln_eps=self.layer_norm_eps,
                                  nsample=self._nsample, knn=self.nneighbor)
            bps, anchor, aidx = hip.load_assets(self._nsample)
            eng = hip.Engine(cfg, w, bps, anchor, aidx, torch.zeros(self.nquery, 3), device)
            if self.nneighbor_query != self.nneighbor:
                eng.set_option("knn_query", self.nneighbor_query)
            self._own_engine = (sig, eng)
        return self._own_engine[1]

    def forward(self, query_xyz, query_feat, pt_xyz, pt_feats):
        """-> (xyz_stack (N_BLOCKS,B,799,3) normalised, pred_pose (B,48)|None, pred_shape (B,10)|None)"""
        if not query_xyz.is_cuda:
            raise RuntimeError("PtEmbedTRv4 runs on the MI355X HIP path only (no CPU fallback)")
        if self._engine_owner is not None:
            eng = self._engine_owner._engine_for(query_xyz.device)
        else:
            self._nsample = pt_xyz.shape[1]
            eng = self._standalone_engine(query_xyz.device)
        out, pose, shape = eng.decoder_forward(query_xyz.float().contiguous(), query_feat.float().contiguous(),
                                               pt_xyz.float().contiguous(), pt_feats.float().contiguous())
        if self.parametric_output:
            # get_parametric_output (pt_metro_transformer.py:139-151 upstream): the last block's coordinates are REPLACED by
            # the MANO layer's output for the regressed (pose, betas) -- rows 21.. the vertices, rows 0..20 the joints
            mano = self.mano_layer if self.mano_layer is not None else getattr(self._engine_owner, "mano_layer", None)
            attached = getattr(eng, "_mano", None)      # the owner head's ManoLayer runs inside the forward: already in `out`
            if attached is not None and getattr(mano, "th_table", None) is attached:
                return out, pose, shape
            if mano is None:
                raise RuntimeError("PARAMETRIC_OUTPUT needs a MANO layer: set_mano_layer(poem_v2_amd.ManoLayer(assets)) "
                                   "(the MANO assets are licence-gated and never read from disk here)")
            m = mano(pose, shape)
            verts, joints = (m.verts, m.joints) if hasattr(m, "verts") else m
            out[-1, :, 21:] = verts
            out[-1, :, :21] = joints
        return out, pose, shape

    mano_layer = None

    def set_mano_layer(self, fn):
        """callable (pose_aa (B,48), betas (B,10)) -> object with .verts (B,778,3) / .joints (B,21,3) or that pair."""
        self.mano_layer = fn
        return self
