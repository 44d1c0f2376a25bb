"""``POEM_Generalized_Head`` -- MI355X-native drop-in for the reference head of the same name
(lib/models/heads/ptEmb_head.py:683-964): same registry name, constructor config keys, forward signature
``forward(mlvl_feat, img_metas, reference_joints, **kwargs) -> {"all_coords_preds", ["pred_pose", "pred_shape"]}``,
``num_preds`` attribute, side effect on ``img_metas["inp_res"]`` and the same ``state_dict`` key names for every
tensor the forward reads.  The nn.Modules below only hold parameters; all arithmetic runs in libpoem_hip.so through
one ``poem_head_forward`` call.  There is no CPU / eager fallback."""
import warnings
import weakref

import numpy as np
import torch
import torch.nn as nn

from . import hip
from .builder import HEAD, build_transformer
from .inputs import synthetic_template
from .weights import live_key_shapes

# Epoch of the process's module TREES: bumped whenever any nn.Module registers a submodule (`head.transformer = ...`,
# `blocks[1] = ...`, parametrize.register_parametrization, add_module).  The head caches the `_parameters` dicts of its
# submodules (see _engine_for) and re-walks its tree only when this moved -- a replaced SUBMODULE is seen at the next forward.
_TREE_EPOCH = [0]
# ... of the module trees that BELONG to a POEM head (weak: a head that is dropped takes its entries along): the hook is
# process-global, and an unrelated model that is being built or edited per request must not force this head's 0.3 ms re-walk.
_TRACKED = weakref.WeakSet()


def _on_module_registration(module, name, submodule):
    if module in _TRACKED:
        _TREE_EPOCH[0] += 1
    return None


torch.nn.modules.module.register_module_module_registration_hook(_on_module_registration)


@HEAD.register_module()
class POEM_Generalized_Head(nn.Module):

    def __init__(self, cfg):
        super().__init__()
        self.nsample = cfg.N_SAMPLE
        self.radius = cfg.RADIUS_SAMPLE
        self.pt_feat_dim = cfg.POINTS_FEAT_DIM
        self.merge_mode = cfg.get("CAM_FEAT_MERGE", "attn")
        self.query_type = cfg.get("QUERY_TYPE", "POEM")
        self.PETR_embedding = cfg.get("PETR_EMBEDDING", False)
        self.parametric_output = cfg.TRANSFORMER.get("PARAMETRIC_OUTPUT", False)
        self.transformer_center_idx = cfg.TRANSFORMER.get("TRANSFORMER_CENTER_IDX", 9)
        self.cfg_transformer = cfg.TRANSFORMER
        self.cfg_position_encoding = cfg.POSITIONAL_ENCODING
        self.num_query = cfg.NUM_QUERY
        self.embed_dims = cfg.EMBED_DIMS
        self.in_channels = cfg.IN_CHANNELS
        self.num_preds = cfg.NUM_PREDS
        self.center_shift = cfg.get("CENTER_SHIFT", False)
        assert self.query_type == "KPT"                                  # ptEmb_head.py:721
        # ptEmb_head.py:65-70: the camera-frustum grid of the PETR embedding (read unconditionally upstream; used when PETR_EMBEDDING)
        self.depth_num = int(cfg.get("DEPTH_NUM", 32))
        self.position_range = [float(v) for v in cfg.get("POSITION_RANGE", [-0.6, -0.6, 0.0, 0.6, 0.6, 1.2])]
        self.LID = bool(cfg.get("LID", False))
        self.depth_start = float(cfg.get("DEPTH_START", 0.0))
        self.depth_end = float(cfg.get("DEPTH_END", 1.2))
        self.pe_normalize = bool(self.cfg_position_encoding.get("NORMALIZE", True))
        if self.PETR_embedding and (3 * self.depth_num) % 8:
            raise ValueError("PETR_EMBEDDING: DEPTH_NUM must be a multiple of 8 (the 3 * DEPTH_NUM input channels of "
                             "position_encoder feed 8-deep matrix-core fragments)")
        if self.cfg_position_encoding.NUM_FEATS * 2 != self.embed_dims:
            raise ValueError("POSITIONAL_ENCODING.NUM_FEATS must be EMBED_DIMS / 2")
        C = self.embed_dims
        # parameter containers with the reference's names (ptEmb_head.py:94,101,701-707,729)
        self.input_proj = nn.Conv2d(self.in_channels, C, kernel_size=1)
        self.adapt_pos3d = nn.Conv2d(C * 3 // 2, C, kernel_size=1)
        if self.PETR_embedding:                # :101-105 upstream (always built there; live only with PETR_EMBEDDING)
            self.position_encoder = nn.Sequential(nn.Conv2d(3 * self.depth_num, C * 2, kernel_size=1), nn.ReLU(),
                                                  nn.Conv2d(C * 2, C, kernel_size=1))
        self.merge_net_feature = nn.ModuleList([
            nn.Sequential(nn.Linear(C, C), nn.ReLU(), nn.Linear(C, C // 2)),
            nn.Sequential(nn.Linear(C // 2, C // 2), nn.ReLU(), nn.Linear(C // 2, C))])
        self.query_feat_embedding = nn.Embedding(799, self.pt_feat_dim)
        self.transformer = build_transformer(self.cfg_transformer)
        self.transformer._engine_owner = None
        object.__setattr__(self.transformer, "_engine_owner", self)      # plain attribute: no module cycle
        # zero-pose template (ManoLayer output upstream, ptEmb_head.py:886-892).  MANO assets are licence-gated:
        # a seeded synthetic template is used until ``set_template`` is called with the real one.
        self.register_buffer("template", synthetic_template(), persistent=False)
        self._template_is_synthetic = True
        self.mano_layer = None    # callable (pose_aa (B,48), betas (B,10)) -> (verts (B,778,3), joints (B,21,3))
        self.max_views = int(cfg.get("MAX_VIEWS", 10))
        self._engine = None          # the engine of the most recent forward
        self._engines = {}           # stream handle -> engine (see _engine_for)
        self._options = {}
        self._engine_sig = None

    # ---- configuration of external inputs -------------------------------------------------------------------
    def set_template(self, template_xyz):
        """(799,3) metres, rows 0..20 joints then 778 vertices, centred at joint TRANSFORMER_CENTER_IDX."""
        t = torch.as_tensor(template_xyz, dtype=torch.float32).reshape(799, 3)
        self.template = t.to(self.template.device)
        self._template_is_synthetic = False
        self._drop_engines()

    def set_mano_layer(self, fn):
        """callable (pose_aa (B,48), betas (B,10)) -> .verts / .joints.  The package's own :class:`~poem_v2_amd.mano.ManoLayer`
        is ATTACHED to the engine instead of called: Q3 -> Linears -> rot6d -> LBS -> last layer all run inside the forward's
        launch graph (include/poem_hip.h poem_attach_mano); any other callable runs between the forward and
        ``poem_finalize_parametric`` as before."""
        self.mano_layer = fn
        for eng in self._engines.values():
            self._attach_mano(eng)

    def _attach_mano(self, eng):
        from .mano import ManoLayer
        m = self.mano_layer
        own = (isinstance(m, ManoLayer) and self.parametric_output and m.th_table.device == eng.device
               and not getattr(self, "mano_in_python", False))
        eng.attach_mano(m.th_table if own else None, m.center_idx if own else 9)
        return own

    def load_reference_state_dict(self, sd):
        """Load a reference checkpoint (full model or head-only); dead tensors are ignored."""
        from .weights import split_state_dict
        live, ignored = split_state_dict(sd, self.embed_dims, in_channels=self.in_channels,
                                         nblocks=self.transformer.layer_num, parametric=self.parametric_output,
                                         petr=self.PETR_embedding, depth_num=self.depth_num)
        missing, unexpected = self.load_state_dict(live, strict=False)
        assert not unexpected, unexpected
        self._drop_engines()
        return ignored

    # ---- engine ------------------------------------------------------------------------------------------------
    def _drop_engines(self):
        self._engines.clear()
        self._engine = None

    def _live_weights(self):
        """The tensors the forward reads, by the reference's key names, as the MODULES present them (attribute access, so a
        parametrized weight -- torch.nn.utils.parametrize -- arrives as its current value, not as `parametrizations.*.original`)."""
        shapes = live_key_shapes(self.embed_dims, self.in_channels, 799, self.transformer.layer_num,
                                 self.parametric_output, petr=self.PETR_embedding, depth_num=self.depth_num)
        out = {}
        with torch.no_grad():
            for k, s in shapes.items():
                obj = self
                for part in k.split("."):
                    obj = obj[int(part)] if part.isdigit() and not hasattr(obj, part) else getattr(obj, part)
                out[k] = obj.detach().reshape(s)
        return out

    def _apply(self, fn, *a, **k):
        self._plist = None                     # .to() / .cuda() / .float(): the parameters move
        return super()._apply(fn, *a, **k)

    def __delattr__(self, name):
        self._plist = None                     # `del head.x`: a removed submodule's cached _parameters dict must not be read again
        _TREE_EPOCH[0] += 1
        super().__delattr__(name)

    def _engine_for(self, device):
        # The engine packs the weights once; it is rebuilt when any parameter's identity, storage or version counter changes
        # (load_state_dict incl. assign=True, `module.weight = nn.Parameter(...)`, .to(), in-place edits).  The submodules'
        # `_parameters` dicts are cached and read directly: walking the module tree for its 199 parameters costs ~0.3 ms of
        # host time per forward -- more than enqueueing the whole step (a hipGraph replay); reading the dicts costs ~0.06 ms
        # and, unlike a cached list of Parameter objects, sees a Parameter that was REPLACED; a replaced SUBMODULE moves
        # _TREE_EPOCH (module-registration hook above) and the list is rebuilt.
        #   ONE ENGINE PER STREAM: an engine's workspace, layout arrays and side streams serve one forward at a time, in stream
        # order.  A caller that alternates small batches over two (or more) torch streams -- the way to fill 256 CUs with batches
        # of 2, the reference's evaluation batch -- gets one engine per stream, so consecutive forwards never share scratch
        # memory and may overlap on the GPU (bench.py small_batch_scope `two_streams`: +7 % at batch 4, nothing at batch <= 2
        # with the default four hardware queues).  Same kernels, same bits.
        if getattr(self, "_plist", None) is None or self._plist_epoch != _TREE_EPOCH[0]:
            mods = list(self.modules())
            self._plist = [m._parameters for m in mods]      # (the module set itself: re-walked when a module tree changed)
            for m in mods:
                _TRACKED.add(m)
            self._plist_epoch = _TREE_EPOCH[0]
        sig = (str(device),) + tuple((id(p), p.data_ptr(), p._version) for d in self._plist for p in d.values() if p is not None)
        if self._engine_sig != sig:
            self._engines.clear()
            self._engine = None
            self._engine_sig = sig
        skey = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
        eng = self._engines.get(skey)
        if eng is None:
            if len(self._engines) >= self.MAX_STREAM_ENGINES:
                self._engines.pop(next(iter(self._engines)))
            t = self.transformer
            cfg = hip.make_config(self.embed_dims, in_channels=self.in_channels, nsample=self.nsample, nquery=799,
                                  heads=t.num_attention_heads, nblocks=t.layer_num, knn=t.nneighbor,
                                  parametric=self.parametric_output, max_views=self.max_views, radius=self.radius, ln_eps=t.layer_norm_eps,
                                  feat_h=self._feat_hw[0], feat_w=self._feat_hw[1], pe_normalize=self.pe_normalize,
                                  petr_embedding=self.PETR_embedding, depth_num=self.depth_num, lid=self.LID,
                                  depth_start=self.depth_start, depth_end=self.depth_end, position_range=self.position_range)
            bps, anchor, aidx = hip.load_assets(self.nsample)
            eng = hip.Engine(cfg, self._live_weights(), bps, anchor, aidx, self.template, device)
            self._engines[skey] = eng
            if self._precision != "fp32":
                eng.set_precision(self._precision)
            if not self._anchor_tables:
                eng.set_anchor_tables(False)
            if not self._chains:
                eng.set_chains(False)
            if t.nneighbor_query != t.nneighbor:
                eng.set_option("knn_query", t.nneighbor_query)
            for k, v in self._options.items():
                eng.set_option(k, v)
            if self.parametric_output:
                self._attach_mano(eng)
        self._engine = eng
        return eng

    MAX_STREAM_ENGINES = 4

    def set_option(self, name, value):
        """A/B switch of the library by name (include/poem_hip.h poem_set_option) on every engine of this head."""
        self._options[name] = int(value)
        for eng in self._engines.values():
            eng.set_option(name, int(value))
        return self

    _anchor_tables = True

    def set_anchor_tables(self, flag=True):
        """Block 0 (fixed anchors, template queries: ptEmb_head.py:886-894, point_transformers.py:10-32): positional products
        of both vector attentions once per forward (default) or, ``False``, per sample exactly as the reference evaluates
        them (include/poem_hip.h poem_set_anchor_tables)."""
        self._anchor_tables = bool(flag)
        for eng in self._engines.values():
            eng.set_anchor_tables(flag)
        return self

    _chains = True

    def set_chains(self, flag=True):
        """Query-side Linears / residuals / LayerNorms of a block as LDS-resident row-tile chains (default) or one launch per
        operator (include/poem_hip.h poem_set_chains)."""
        self._chains = bool(flag)
        for eng in self._engines.values():
            eng.set_chains(flag)
        return self

    _precision = "fp32"

    def set_precision(self, mode):
        """``"fp32"`` (default: exact fp32 matrix-core products everywhere) or ``"split_f16x3"`` (opt-in: the three C x C
        per-neighbour GEMMs of the vector attention as hi/lo f16 splits with fp32 accumulation, include/poem_hip.h)."""
        if mode not in hip.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(hip.PRECISIONS)}")
        self._precision = mode
        for eng in self._engines.values():
            eng.set_precision(mode)
        return self

    _feat_hw = (16, 16)

    # ---- forward -----------------------------------------------------------------------------------------------
    def forward(self, mlvl_feat, img_metas, reference_joints, **kwargs):
        # Inference only.  The reference trains through this head (scripts/train_ddp.py:84, lib/models/POEM.py:363-466
        # upstream: losses on all_coords_preds / pred_pose / pred_shape back-propagate into the head and HRNet); the HIP
        # path has no backward and returns tensors without a graph, so a training step would silently receive no gradient.
        # Refuse instead: train mode with grad enabled, or any input that asks for a gradient.
        if torch.is_grad_enabled() and (self.training or mlvl_feat.requires_grad or reference_joints.requires_grad):
            raise RuntimeError("POEM_Generalized_Head (HIP) is inference-only: no backward is built.  Call model.eval() and run under "
                               "torch.no_grad() (as scripts/eval.py does); for training keep the reference's PyTorch head")
        if not mlvl_feat.is_cuda:
            raise RuntimeError("POEM_Generalized_Head runs on the MI355X HIP path only (no CPU fallback)")
        assert self.merge_mode == "attn"                                                   # ptEmb_head.py:903
        assert int(np.sum(np.asarray(img_metas["master_id"]))) == 0, "only support master_id is 0"   # :750-751
        device = mlvl_feat.device
        inp_img_w, inp_img_h = img_metas["inp_img_shape"]                                 # upstream naming, :831
        # :832-833 upstream.  Cached per (shape, device): building a device tensor from a Python list is a pageable H2D copy,
        # which blocks the host until the stream reaches it -- i.e. until the previous forward has finished (the host
        # could then never run ahead of the GPU: ~0.5-1 ms of idle GPU between consecutive forwards).
        key = (int(inp_img_w), int(inp_img_h), str(device))
        cache = getattr(self, "_inp_res_cache", None)
        if cache is None or cache[0] != key:
            cache = (key, torch.tensor([inp_img_w, inp_img_h], dtype=torch.float32, device=device))
            self._inp_res_cache = cache
        img_metas["inp_res"] = cache[1]
        assert mlvl_feat.shape[1] == self.in_channels
        if tuple(mlvl_feat.shape[-2:]) != self._feat_hw:
            self._feat_hw = tuple(int(v) for v in mlvl_feat.shape[-2:])
            self._drop_engines()
        views = np.asarray(img_metas["cam_view_num"]).astype(np.int64)
        if views.max() > self.max_views:
            self.max_views = int(views.max())
            self._drop_engines()
        if self._template_is_synthetic and not getattr(self, "_warned", False):
            warnings.warn("POEM_Generalized_Head: using the synthetic hand template (MANO assets absent); "
                          "call set_template() with ManoLayer's zero-pose output for real checkpoints")
            self._warned = True
        eng = self._engine_for(device)
        f32 = lambda t: t.to(device=device, dtype=torch.float32).contiguous()   # noqa: E731
        out, pose, betas = eng.head_forward(f32(mlvl_feat), f32(img_metas["cam_intr"]), f32(img_metas["cam_extr"]), views,
                                            f32(reference_joints), (inp_img_w, inp_img_h))
        results = {"all_coords_preds": out}
        if self.parametric_output:
            if self.mano_layer is None:
                raise RuntimeError("PARAMETRIC_OUTPUT needs a MANO layer: call set_mano_layer(fn)")
            if getattr(eng, "_mano", None) is None:      # (an attached ManoLayer has already run inside the forward)
                m = self.mano_layer(pose, betas)
                verts, joints = (m.verts, m.joints) if hasattr(m, "verts") else m
                eng.finalize_parametric(f32(verts), f32(joints), f32(reference_joints), out)
            results["pred_pose"] = pose.reshape(-1, 16, 3)
            results["pred_shape"] = betas.reshape(-1, 10)
        return results
