"""Import harness for the upstream reference (this container only).

TEST INFRASTRUCTURE.  Imports ``/root/reference/lib`` *by path* (nothing is copied) after seeding
``sys.modules`` with small stand-ins for the third-party packages this image lacks, so that the
reference's own ``POEM_Generalized_Head`` / ``PtEmbedTRv4`` can be built and run on CPU to produce
golden vectors (``make_golden.py``).  ``/root/reference`` does not exist on the GPU box: nothing in
the product, ``bench.py``, ``smoke()`` or the ``-m gpu`` tests imports this file.

Stand-ins:
  * ``yacs.config.CfgNode``      dict + attribute access (clone/defrost/freeze/merge...)
  * ``pytorch3d.ops.knn_points`` direct squared-L2 (sum of (a-b)^2, no matmul trick) + ``topk`` sorted
                                 ascending -- the published semantics of pytorch3d 0.7.x ``knn_points``
                                 (tie order is implementation-defined upstream: parity unpinned there)
  * ``manotorch.manolayer.ManoLayer``  returns a fixed, seeded synthetic 21-joint/778-vertex template
                                 (MANO assets are licence-gated and absent).  The template values are
                                 an *input* of the path, fed identically to oracle and HIP.
  * transformers v5 -> v4 shim   ``BertAttention`` with v4 ctor/forward semantics (cross-attention when
                                 ``encoder_hidden_states`` is passed) and v4 ``init_weights``.
  * MagicMock for cv2, imageio, git, torchvision, webdataset, open3d, ...
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

REF_ROOT = "/root/reference"
sys.dont_write_bytecode = True


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "oracle"))
from poem_oracle import synthetic_template, toy_mano  # noqa: E402  (shared stand-ins; both test infrastructure)


class _ManoOut:
    def __init__(self, verts, joints):
        self.verts = verts
        self.joints = joints


class _FakeManoLayer(torch.nn.Module):
    """Zero-pose template provider; parametric calls (medium_MANO tail) return a deterministic,
    differentiable-free function of (pose, betas) so that the Q3 plumbing can be exercised."""

    def __init__(self, *a, center_idx=None, **k):
        super().__init__()
        self.center_idx = center_idx
        t = synthetic_template()
        self.register_buffer("tmpl", t, persistent=False)
        self.th_faces = torch.zeros(1538, 3, dtype=torch.long)
        self.th_J_regressor = torch.zeros(16, 778)

    def forward(self, pose, betas, **k):
        verts, joints = toy_mano(self.tmpl.to(pose.device), self.center_idx)(pose, betas)
        return _ManoOut(verts, joints)

    def get_mano_closed_faces(self):
        return self.th_faces


class _CfgNode(dict):
    """Minimal yacs.config.CfgNode stand-in."""

    def __init__(self, init_dict=None, key_list=None, new_allowed=False):
        super().__init__()
        init_dict = {} if init_dict is None else init_dict
        for k, v in init_dict.items():
            if isinstance(v, dict) and not isinstance(v, _CfgNode):
                v = type(self)(v)
            self[k] = v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def clone(self):
        import copy
        return copy.deepcopy(self)

    def defrost(self):
        pass

    def freeze(self):
        pass

    def set_new_allowed(self, v):
        pass

    def is_frozen(self):
        return False

    def merge_from_other_cfg(self, other):
        for k, v in other.items():
            if k in self and isinstance(self[k], dict) and isinstance(v, dict):
                self[k].merge_from_other_cfg(v)
            else:
                self[k] = v

    def merge_from_file(self, path):
        import yaml
        with open(path) as f:
            self.merge_from_other_cfg(type(self)(yaml.safe_load(f)))

    def dump(self, *a, **k):
        import yaml
        return yaml.safe_dump(dict(self))


KNN_FMA = False      # make_golden.py sets it per case ("*_fma": the distance rounding of pytorch3d's CUDA kernel)


def knn_points_direct(p1, p2, K, return_nn=False, **kw):
    """pytorch3d.ops.knn_points stand-in: the accumulation loop of pytorch3d's kernels (poem_oracle.knn_distances --
    ((dx*dx + dy*dy) + dz*dz) as knn_cpu.cpp rounds it, or with KNN_FMA the fma-contracted form nvcc makes of knn.cu),
    K smallest, ascending."""
    import poem_oracle as po
    dist = po.knn_distances(p1, p2, fma=KNN_FMA)
    val, idx = torch.topk(dist, K, dim=-1, largest=False, sorted=True)
    nn = None
    if return_nn:
        nn = torch.gather(p2[:, None].expand(-1, p1.shape[1], -1, -1), 2, idx[..., None].expand(-1, -1, -1, 3))
    return val, idx, nn


def _install_stubs():
    import transformers  # noqa: F401  must be imported before torchvision is mocked
    import transformers.models.bert.modeling_bert as mb

    yacs = types.ModuleType("yacs")
    yacs_cfg = types.ModuleType("yacs.config")
    yacs_cfg.CfgNode = _CfgNode
    yacs.config = yacs_cfg
    sys.modules["yacs"] = yacs
    sys.modules["yacs.config"] = yacs_cfg

    tc = types.ModuleType("termcolor")
    tc.colored = lambda s, *a, **k: s
    tc.cprint = lambda *a, **k: None
    sys.modules["termcolor"] = tc

    p3d = types.ModuleType("pytorch3d")
    p3d_ops = types.ModuleType("pytorch3d.ops")
    p3d_ops.knn_points = knn_points_direct
    p3d_ops.ball_query = MagicMock()
    p3d_ops.sample_farthest_points = MagicMock()
    p3d.ops = p3d_ops
    sys.modules["pytorch3d"] = p3d
    sys.modules["pytorch3d.ops"] = p3d_ops
    for sub in ["renderer", "structures", "io", "renderer.mesh", "utils", "loss"]:
        sys.modules["pytorch3d." + sub] = MagicMock()
    # pytorch3d.transforms: the three functions the medium_MANO tail reaches are the oracle's restatements of the
    # published algorithms (third-party source absent -> parity UNPINNED for them); the rest are never called.
    import poem_oracle as po
    p3d_tr = types.ModuleType("pytorch3d.transforms")
    p3d_tr.rotation_6d_to_matrix = po.rotation_6d_to_matrix
    p3d_tr.matrix_to_quaternion = po.matrix_to_quaternion
    p3d_tr.quaternion_to_axis_angle = po.quaternion_to_axis_angle

    def _absent(name):
        def f(*a, **k):
            raise NotImplementedError(f"pytorch3d.transforms.{name} is not available in this container")
        f.__name__ = name
        return f

    for n in ["axis_angle_to_matrix", "axis_angle_to_quaternion", "euler_angles_to_matrix", "matrix_to_euler_angles",
              "matrix_to_rotation_6d", "quaternion_to_matrix"]:
        setattr(p3d_tr, n, _absent(n))
    p3d.transforms = p3d_tr
    sys.modules["pytorch3d.transforms"] = p3d_tr

    mt = types.ModuleType("manotorch")
    mtl = types.ModuleType("manotorch.manolayer")
    mtl.ManoLayer = _FakeManoLayer
    mtl.MANOOutput = _ManoOut
    mt.manolayer = mtl
    sys.modules["manotorch"] = mt
    sys.modules["manotorch.manolayer"] = mtl
    for sub in ["axislayer", "anchorlayer", "utils", "utils.quatutils", "utils.geometry", "upsamplelayer",
                "anatomy_loss"]:
        sys.modules["manotorch." + sub] = MagicMock()

    for name in ["cv2", "imageio", "git", "torchvision", "torchvision.transforms", "torchvision.transforms.functional",
                 "torchvision.models", "torchvision.ops", "torchvision.utils", "webdataset", "open3d", "trimesh",
                 "chumpy", "opendr", "opendr.renderer", "opendr.camera", "opendr.lighting", "pyrender",
                 "neural_renderer", "tensorboard", "torch.utils.tensorboard", "deprecated", "prettytable",
                 "dex_ycb_toolkit", "dex_ycb_toolkit.factory", "dex_ycb_toolkit.dex_ycb", "pycocotools",
                 "pycocotools.coco", "skimage", "skimage.io", "matplotlib", "matplotlib.pyplot", "PIL", "PIL.Image",
                 "tqdm.contrib", "roma", "smplx", "json_tricks", "pyquaternion", "oikit", "oikit.oi_image",
                 "mpl_toolkits", "mpl_toolkits.mplot3d", "kornia", "kornia.geometry", "kornia.geometry.transform"]:
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = MagicMock()

    # --- transformers v5 -> v4 semantics ------------------------------------------------------------
    _V5Attn = mb.BertAttention

    class BertAttentionV4(_V5Attn):
        def __init__(self, config, position_embedding_type=None):
            try:
                super().__init__(config, is_cross_attention=True)
            except TypeError:  # genuine v4
                super().__init__(config, position_embedding_type=position_embedding_type)

        def forward(self, hidden_states, attention_mask=None, head_mask=None, encoder_hidden_states=None,
                    encoder_attention_mask=None, past_key_value=None, output_attentions=False, **kw):
            out = super().forward(hidden_states, attention_mask=None, encoder_hidden_states=encoder_hidden_states,
                                  encoder_attention_mask=None)
            return out if isinstance(out, tuple) else (out,)

    mb.BertAttention = BertAttentionV4

    def _v4_init_weights(self):
        std = self.config.initializer_range
        for m in self.modules():
            if isinstance(m, (torch.nn.Linear, torch.nn.Embedding)):
                m.weight.data.normal_(mean=0.0, std=std)
                if isinstance(m, torch.nn.Linear) and m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, torch.nn.LayerNorm):
                m.bias.data.zero_()
                m.weight.data.fill_(1.0)

    mb.BertPreTrainedModel.init_weights = _v4_init_weights
    if not hasattr(mb, "apply_chunking_to_forward"):
        from transformers.pytorch_utils import apply_chunking_to_forward
        mb.apply_chunking_to_forward = apply_chunking_to_forward


_READY = False


def setup(cwd=None):
    """Install stubs, chdir to the reference root (relative asset paths) and import the reference."""
    global _READY
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError("reference tree not present; golden vectors can only be (re)generated in the build container")
    if not _READY:
        _install_stubs()
        if REF_ROOT not in sys.path:
            sys.path.insert(0, REF_ROOT)
        _READY = True
    os.chdir(cwd or REF_ROOT)
    import lib.models  # noqa: F401  triggers registry population
    from lib.utils.config import CN
    from lib.models.heads import build_head
    return CN, build_head


def load_head_cfg(CN, model="medium"):
    import yaml
    name = {"small": "train_small", "medium": "train_medium", "large": "train_large", "huge": "train_huge",
            "medium_MANO": "train_medium_MANO"}[model]
    with open(os.path.join(REF_ROOT, "config/release", name + ".yaml")) as f:
        y = yaml.safe_load(f)
    return CN(y["MODEL"]["HEAD"]), y
