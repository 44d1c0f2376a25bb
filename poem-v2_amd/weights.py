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


def live_key_shapes(embed, in_channels=160, nquery=799, nblocks=3, parametric=False):
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
    return k


def seeded_state_dict(embed, seed=0, gain=1.0, **kw):
    """Deterministic, well-conditioned weights: every tensor is drawn from its own CPU generator seeded by
    (seed, crc32(key)), so the reference module and this build can be filled identically without shipping blobs.

    Scales follow the reference's initialisers: N(0, 0.02) for every Linear inside a decoder block (BERT v4
    ``init_weights``, applied by point_METRO_block to all sub-modules), U(+-1/sqrt(fan_in)) for head-level
    Linear/Conv, N(0,1) for the query embedding; biases and LayerNorm offsets get small non-zero values so that
    every term of the arithmetic is exercised.  ``gain`` scales the block weights ("hot" stress variant)."""
    out = OrderedDict()
    for key, shape in live_key_shapes(embed, **kw).items():
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(key.encode())) & 0x7FFFFFFF)
        in_block = key.startswith("transformer.")
        if key.endswith("LayerNorm.weight"):
            t = 1.0 + 0.02 * torch.randn(shape, generator=g)
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
