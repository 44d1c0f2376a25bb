"""Checkpoint surface of the path: the reference's ``state_dict`` key names (relative to the head, i.e. without the
``ptEmb_head.`` prefix of the full-model checkpoint) for the tensors the forward actually reads, their shapes,
and a seeded generator used wherever real checkpoints are unavailable (bench, tests, golden vectors).

Key names follow SURVEY.md section 8b / the reference modules:
  ptEmb_head.py:94,101,701-707,729  (input_proj, adapt_pos3d, merge_net_feature, query_feat_embedding)
  pt_metro_transformer.py:25-31,49-54,113,125-126  (reg_branch, attn/cross_attn, embedding, flat_verts, mano_linear)
  point_transformers.py:49-56,101-108  (fc1, fc2, fc_delta, fc_gamma, w_qs, w_ks, w_vs)
Dead tensors the reference also serialises (center_shift_layer, reg_branches, position_encoder, BERT word
embeddings, pooler, ...) are accepted and ignored by ``split_state_dict``."""
import zlib
from collections import OrderedDict

import torch

MODEL_EMBED = {"small": 128, "medium": 256, "large": 512, "huge": 1024, "medium_MANO": 256}


def live_key_shapes(embed, in_channels=160, nquery=799, nblocks=3, parametric=False, petr=False, depth_num=32):
    C = embed
    k = OrderedDict()

    def lin(name, o, i, bias=True):
        k[name + ".weight"] = (o, i)
        if bias:
            k[name + ".bias"] = (o,)

    k["input_proj.weight"] = (C, in_channels, 1, 1)
    k["input_proj.bias"] = (C,)
    k["adapt_pos3d.weight"] = (C, 3 * C // 2, 1, 1)
    k["adapt_pos3d.bias"] = (C,)
    lin("merge_net_feature.0.0", C, C)
    lin("merge_net_feature.0.2", C // 2, C)
    lin("merge_net_feature.1.0", C // 2, C // 2)
    lin("merge_net_feature.1.2", C, C // 2)
    k["query_feat_embedding.weight"] = (nquery, C)
    for i in range(nblocks):
        p = f"transformer.pt_metro_encoder.{i}."
        lin(p + "embedding", C, C)
        for a in ("attn", "cross_attn"):
            for n in ("query", "key", "value"):
                lin(p + f"encoder.{a}.self.{n}", C, C)
            lin(p + f"encoder.{a}.output.dense", C, C)
            k[p + f"encoder.{a}.output.LayerNorm.weight"] = (C,)
            k[p + f"encoder.{a}.output.LayerNorm.bias"] = (C,)
        for a in ("query_self_attn", "query_cross_attn"):
            q = p + f"encoder.vec_attn.{a}."
            lin(q + "fc1", C, C)
            lin(q + "fc2", C, C)
            lin(q + "fc_delta.0", C, 3)
            lin(q + "fc_delta.2", C, C)
            lin(q + "fc_gamma.0", C, C)
            lin(q + "fc_gamma.2", C, C)
            lin(q + "w_qs", C, C, bias=False)
            lin(q + "w_ks", C, C, bias=False)
            lin(q + "w_vs", C, C, bias=False)
        lin(p + "encoder.vec_attn.reg_branch.0", C, C)
        lin(p + "encoder.vec_attn.reg_branch.2", 3, C)
        lin(p + "encoder.intermediate.dense", 4 * C, C)
        lin(p + "encoder.output.dense", C, 4 * C)
        k[p + "encoder.output.LayerNorm.weight"] = (C,)
        k[p + "encoder.output.LayerNorm.bias"] = (C,)
        if parametric:
            lin(p + "flat_verts", 1, nquery)
            lin(p + "mano_linear", 106, C)
    if petr:
        # PETR_EMBEDDING=True (ptEmb_head.py:101-105,865-867): position_encoder is live.  Listed LAST (after the blocks, not at
        # its state_dict position behind adapt_pos3d) so that the canonical indices of every other tensor do not depend on it.
        k["position_encoder.0.weight"] = (2 * C, 3 * depth_num, 1, 1)
        k["position_encoder.0.bias"] = (2 * C,)
        k["position_encoder.2.weight"] = (C, 2 * C, 1, 1)
        k["position_encoder.2.bias"] = (C,)
    return k


def seeded_state_dict(embed, seed=0, gain=1.0, ln_spread=0.02, **kw):
    """Deterministic, well-conditioned weights: every tensor is drawn from its own CPU generator seeded by
    (seed, crc32(key)), so the reference module and this build can be filled identically without shipping blobs.

    Scales follow the reference's initialisers: N(0, 0.02) for every Linear inside a decoder block (BERT v4
    ``init_weights``, applied by point_METRO_block to all sub-modules), U(+-1/sqrt(fan_in)) for head-level
    Linear/Conv, N(0,1) for the query embedding; biases and LayerNorm offsets get small non-zero values so that
    every term of the arithmetic is exercised.  ``gain`` scales the block weights and ``ln_spread`` the deviation of the
    LayerNorm gains from 1 ("hot" stress variant: with gain 1 the coordinate update of a block is ~1e-3 normalised units, so
    attention-path error is attenuated ~1000x before it reaches a vertex and the neighbour sets of blocks 1, 2 are
    essentially the template's; gain 2.5 makes activations, updates and neighbour changes O(1) -- the conditioning of a
    trained checkpoint, tests/golden/*_hot.npz)."""
    out = OrderedDict()
    for key, shape in live_key_shapes(embed, **kw).items():
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFF)
        in_block = key.startswith("transformer.")
        if key.endswith("LayerNorm.weight"):
            t = 1.0 + ln_spread * torch.randn(shape, generator=g)
        elif key.endswith("LayerNorm.bias"):
            t = 0.02 * torch.randn(shape, generator=g)
        elif key == "query_feat_embedding.weight":
            t = torch.randn(shape, generator=g)
        elif in_block:
            t = 0.02 * gain * torch.randn(shape, generator=g)
            if "fc_delta.0.weight" in key:       # 3 -> C layer acts on O(1) coordinates
                t = t * 10.0
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            bound = 1.0 / (fan_in ** 0.5) if len(shape) > 1 else 0.05
            t = (torch.rand(shape, generator=g) * 2 - 1) * bound
        out[key] = t.float().contiguous()
    return out


def split_state_dict(sd, embed, strip_prefixes=("module.", "ptEmb_head."), **kw):
    """Pick the live tensors out of a reference checkpoint ``state_dict`` (full-model or head-only); tolerate dead
    and unexpected keys (the reference loads with strict=True, the dead tensors are simply unused)."""
    want = live_key_shapes(embed, **kw)
    norm = {}
    for key, val in sd.items():
        k = key
        changed = True
        while changed:
            changed = False
            for p in strip_prefixes:
                if k.startswith(p):
                    k = k[len(p):]
                    changed = True
        norm[k] = val
    live, missing = OrderedDict(), []
    for key, shape in want.items():
        if key not in norm:
            missing.append(key)
            continue
        t = torch.as_tensor(norm[key]).float()
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{key}: checkpoint shape {tuple(t.shape)} != expected {tuple(shape)}")
        live[key] = t.contiguous()
    if missing:
        raise KeyError(f"checkpoint lacks {len(missing)} live tensors, e.g. {missing[:4]}")
    ignored = sorted(set(norm) - set(want))
    return live, ignored


# ---- convolutional glue in front of the head (poem_v2_amd.decode) ----------------------------------
_FEAT_SIZE = (40, 80, 160, 320)          # lib/models/POEM.py:55-56 upstream (HRNet)


def decoder_key_shapes():
    """state_dict keys (relative to the model) and shapes of feat_delayer / feat_in / uv_delayer / uv_out
    (lib/models/POEM.py:84-112 upstream, HRNet branch)."""
    f = _FEAT_SIZE
    ks = {}

    def block(name, cin, cout, k, norm):
        ks[f"{name}.conv.weight"] = (cout, cin, k, k)
        ks[f"{name}.conv.bias"] = (cout,)
        if norm:
            for n in ("weight", "bias", "running_mean", "running_var"):
                ks[f"{name}.norm.{n}"] = (cout,)

    for i in range(3):
        block(f"feat_delayer.{i}", f[i], f[i + 1], 3, True)
    block("feat_in", f[3], f[2], 1, False)
    block("uv_delayer.0", f[3] + f[2], f[2], 3, True)
    block("uv_delayer.1", f[2] + f[1], f[1], 3, True)
    block("uv_delayer.2", f[1] + f[0], f[0], 3, True)
    block("uv_out", f[0], 21, 1, False)
    return ks


def seeded_decoder_state_dict(seed=0):
    """Deterministic weights for fixtures / benches: conv weights N(0, sqrt(2 / fan_out)) as ConvBlock's
    kaiming_normal_(mode='fan_out') draws them (lib/models/bricks/conv.py:31-33 upstream), small random biases, and
    *non-trivial* BatchNorm statistics so that the folded affine is exercised (the reference initialises gamma = 1,
    beta = 0, mean = 0, var = 1)."""
    g = torch.Generator().manual_seed(1000 + seed)
    sd = {}
    for k, shp in decoder_key_shapes().items():
        if k.endswith("conv.weight"):
            fan_out = shp[0] * shp[2] * shp[3]
            sd[k] = torch.randn(shp, generator=g) * (2.0 / fan_out) ** 0.5
        elif k.endswith("conv.bias"):
            sd[k] = 0.05 * torch.randn(shp, generator=g)
        elif k.endswith("norm.weight"):
            sd[k] = 1.0 + 0.2 * torch.randn(shp, generator=g)
        elif k.endswith("norm.bias"):
            sd[k] = 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("running_mean"):
            sd[k] = 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(shp, generator=g)
    return sd
