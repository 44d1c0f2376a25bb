"""Release configurations of the path: the ``MODEL.HEAD`` subtree of the reference's config/release/train_*.yaml
(:185-225) with the four size fields set the way scripts/eval_single.py:38-39,74-86 rewrites them per model."""
from .config import CN
from .weights import MODEL_EMBED


def head_cfg(embed=256, nsample=4096, parametric=False, max_views=10):
    return CN({
        "TYPE": "POEM_Generalized_Head",
        "TRANSFORMER": {"TYPE": "PtEmbedTRv4", "N_BLOCKS": 3, "INPUT_FEAT_DIM": embed, "NUM_HIDDEN_LAYERS": 4,
                        "NUM_ATTENTION_HEADS": 4, "DROPOUT": 0.1, "BPS_FEAT_DIM": nsample, "N_NEIGHBOR": 32,
                        "N_NEIGHBOR_QUERY": 32, "PARAMETRIC_OUTPUT": parametric},
        "POSITIONAL_ENCODING": {"TYPE": "SinePositionalEncoding3D", "NUM_FEATS": embed // 2, "NORMALIZE": True},
        "WITH_POSITION": True, "WITH_MULTIVIEW": True, "NUM_QUERY": 799, "NUM_PREDS": 3, "NUM_REG_FCS": 2,
        "DEPTH_NUM": 32, "POSITION_RANGE": [-0.6, -0.6, 0.0, 0.6, 0.6, 1.2], "LID": False, "DEPTH_START": 0.0,
        "DEPTH_END": 1.2, "POINTS_FEAT_DIM": embed, "EMBED_DIMS": embed, "IN_CHANNELS": 160, "CENTER_SHIFT": True,
        "N_SAMPLE": nsample, "RADIUS_SAMPLE": 0.1, "CAM_FEAT_MERGE": "attn", "QUERY_TYPE": "KPT",
        "MAX_VIEWS": max_views})


def model_head_cfg(model="medium", **kw):
    """model in {small, medium, large, huge, medium_MANO} (scripts/eval_single.py:38-39 upstream)."""
    return head_cfg(MODEL_EMBED[model], parametric=(model == "medium_MANO"), **kw)
